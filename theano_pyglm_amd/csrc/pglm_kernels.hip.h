// Device kernels of the population-GLM likelihood library (gfx950 / CDNA4 only).
//
// Reference arithmetic (slinderman/theano_pyglm):
//   features   fS[t,n',b] = sum_{tau=1..R} S[t-tau,n'] ibasis[tau-1,b]   pyglm/utils/basis.py:201-236
//   currents   x[t,n] = bias_n + fstim[t,:].wstim_n + sum_{n',b} fS[t,n',b] w_n[n',b] Weff[n',n]
//                                                                         pyglm/glm.py:31-45, impulse.py:58
//   likelihood ll_n = sum_t ( -dt*lam + log(lam)*S[t,n] ),  lam = nlin(x)   pyglm/glm.py:52, nlin.py:25,43
//   gradient   T.grad(glm.ll, [bias, w_stim, w_ir])                        pyglm/inference/coord_descent.py:27-30
//
// Design.  The likelihood and its gradient are two dense contractions around an
// elementwise rate epilogue:  X = F.Wmat  (forward), r = dll/dx, G += F^T.r  (backward), all on
// v_mfma_f64_16x16x4_f64; the forward accumulator layout IS the B-operand layout of the backward
// MFMA, so r never leaves its registers in between.  Work is cut into 16-row time tiles; a
// workgroup walks a chunk of tiles with G in registers.  Three kernel families share that scheme:
//   k_build_fimg + k_fused5  F resident in HBM as LDS-shaped tiles (built once per data set, like the
//                            reference's data['fS']) and streamed with LDS-DMA; two passes over the
//                            chunk because G of 128 post neurons (655 KB) exceeds one CU's registers
//   k_fused3                 the same two passes, F tile regenerated in LDS from the spike events
//   k_fused2                 one pass, post tiles x K slices across the 8 waves (small post blocks,
//                            the sliced path for N > 128, optional f32 feature tile)
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
    double* __restrict__ llpart;         // [nChunks][nPT][64]
    double* __restrict__ gbpart;         // [nChunks][nPT][64]
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
};

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
        const size_t slot = ((size_t)chunk * p.nPT + pt) * KSPLIT + ksl;
        p.llpart[slot * 64 + lane] = ll_acc;
        p.gbpart[slot * 64 + lane] = gb_acc;
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
        const size_t slot = (size_t)chunk * p.nPT + pt;
        p.llpart[slot * 64 + lane] = ll_acc;
        p.gbpart[slot * 64 + lane] = gb_acc;
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
// HLP = 1: blocks of FIVE or SIX post tiles (N = 65 .. 96 and the last block of 13 or 14 tiles) leave three or two of the
// eight waves without a tile of their own, and two SIMDs with two tiles each: the idle waves take over part of the work of
// the tiles of a doubly loaded SIMD -- in pass 1 the second half of a tile's forward k-steps (the partial currents reach
// the tile's own wave through LDS, in front of the barrier that closes the forward phase anyway), in pass 2 the second half
// of its k-tiles (own G registers, own partials).  Five tiles: waves 5 / 6 help tiles 0 / 4 (both on SIMD 0); six tiles:
// waves 6 / 7 help tiles 0 / 1.  Per tile step the busiest SIMD then carries 1.5 forward passes instead of 2
// (five tiles: 1 + the two L backward passes).  Tiles without a helper compute exactly what HLP = 0 computes.
template <int KTL, int KTH, int PASS, int XIN = 0, int PART = 0, int HLP = 0>
__global__ __launch_bounds__(512, 2) void k_fused5(const FusedParams p)
{
    static_assert(!HLP || (XIN == 0 && PART == 0 && KTL >= 2 && KTH >= 2), "helper waves: plain two-pass form only");
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
    const int pb = blockIdx.x % nPB;
    const int chunk = blockIdx.x / nPB;
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
        // HLP: partial currents of the two helpers, [2][4][64], behind the spike scratch
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
            const size_t slot = (size_t)chunk * p.nPT + pt;
            p.llpart[slot * 64 + lane] = ll_acc;
            p.gbpart[slot * 64 + lane] = gb_acc;
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
        const size_t slot = ((size_t)chunk * p.nPT + pt) * KSPLIT + ksl;
        p.llpart[slot * 64 + lane] = ll_acc;
        p.gbpart[slot * 64 + lane] = gb_acc;
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

    const size_t slot = ((size_t)chunk * p.nPT + pt) * KSPLIT + ksl;
    p.llpart[slot * 64 + lane] = ll_acc;
    p.gbpart[slot * 64 + lane] = gb_acc;
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
        const size_t slot = (size_t)chunk * p.nPT + pt;
        p.llpart[slot * 64 + lane] = ll_acc;
        p.gbpart[slot * 64 + lane] = gb_acc;
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

// ---------------------------------------------------------------------------
// prep: Wmat in MFMA B-fragment order + bias vector
//   Wfrag[pt][ks][lane] = Wmat[k = 4ks + (lane>>4)][n = 16pt + (lane&15)]
// ---------------------------------------------------------------------------
__global__ void k_prep_w(const double* __restrict__ theta, const double* __restrict__ Weff,
                         double* __restrict__ Wfrag, double* __restrict__ bias, int N, int B,
                         int Dstim, int Kimp, int Ktot, int KS, int n_lo, int npost, int nPT,
                         int pair, int Nall, int np0, int DsAll, int ds0, const int* __restrict__ pidx)
{
    // N / Dstim / Kimp describe the launch's feature-column slice (see FusedParams)
    const int P = 1 + DsAll + Nall * B;
    const long long total = (long long)nPT * KS * 64;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int ks = (int)((i >> 6) % KS);
        const int pt = (int)((i >> 6) / KS);
        const int k = 4 * ks + (lane >> 4);
        const int n = 16 * pt + (lane & 15);
        double v = 0.0;
        if (n < npost && k < Ktot) {
            if (k < Kimp) {
                const int npre = np0 + k / B;
                v = theta[(size_t)n * P + 1 + DsAll + np0 * B + k] * Weff[(size_t)npre * Nall + (pidx ? pidx[n] : n_lo + n)];
            } else {
                v = theta[(size_t)n * P + 1 + ds0 + (k - Kimp)];
            }
        }
        // pair layout (V2): [pt][ks/2][lane][ks&1] so that one 16-byte load feeds two k-steps
        const long long o = pair ? (((long long)pt * (KS / 2) + (ks >> 1)) * 64 + lane) * 2 + (ks & 1) : i;
        Wfrag[o] = v;
    }
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nPT * 16; n += gridDim.x * blockDim.x)
        bias[n] = (n < npost) ? theta[(size_t)n * P] : 0.0;
}

// ---------------------------------------------------------------------------
// finalize: deterministic reduction of the per-chunk partials, Weff chain rule,
// scatter into the (npost, P) gradient layout
// ---------------------------------------------------------------------------
// ll_n and d ll_n / d bias of neuron n: all threads of the block stride over the (chunk, k-slice wave, lane group)
// partials with four independent sums each, fixed-order butterfly + fixed-order combination of the waves --
// deterministic for a given launch geometry.  (One wave walking the partials alone is a serial chain of
// nChunks * nsub / 16 dependent global loads: 150 us for the 1 250 chunks of a 4-neuron population.)
__device__ __forceinline__ void pgl_reduce_ll(const double* __restrict__ llpart, const double* __restrict__ gbpart,
                                              double* __restrict__ ll_out, double* __restrict__ grad_out,
                                              const int n, const int P, const int nPT, const int nChunks,
                                              const int nsub, double (*red)[64])
{
    // always the first 256 threads of the block, whatever its size: the ll of an ll-only call (k_finalize_ll) and of
    // an ll+grad call (trailing blocks of k_finalize) are then the same sums in the same order, bit for bit
    const int nthr = 256, t = (int)threadIdx.x, lane = t & 63, w = t >> 6, nw = 4;
    const int pt = n >> 4, col = n & 15;
    const int per = 4 * nsub;
    const int total = nChunks * per;
    double sl[4] = {0.0, 0.0, 0.0, 0.0}, sg[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = t; i0 < total && t < nthr; i0 += 4 * nthr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * nthr;
            if (i < total) {
                const int c = i / per, g = i - c * per;
                const size_t idx = ((size_t)c * nPT + pt) * nsub * 64 + (size_t)g * 16 + col;
                sl[j] += llpart[idx];
                sg[j] += gbpart[idx];
            }
        }
    }
    double a = (sl[0] + sl[1]) + (sl[2] + sl[3]), b = (sg[0] + sg[1]) + (sg[2] + sg[3]);
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (lane == 0 && w < nw) {
        red[w][0] = a;
        red[w][1] = b;
    }
    __syncthreads();
    if (t == 0) {
        double x = 0.0, y = 0.0;
        for (int j = 0; j < nw; ++j) {
            x += red[j][0];
            y += red[j][1];
        }
        ll_out[n] = x;
        if (grad_out != nullptr) grad_out[(size_t)n * P] = y;
    }
}

__global__ __launch_bounds__(1024) void k_finalize(const double* __restrict__ Gpart, const double* __restrict__ llpart,
                           const double* __restrict__ gbpart, const double* __restrict__ Weff,
                           double* __restrict__ ll_out, double* __restrict__ grad_out, int N, int B,
                           int Dstim, int Kimp, int Ktot, int KT, int n_lo, int npost, int nPT,
                           int nChunks, int Nall, int np0, int DsAll, int ds0, int nsub, int kt0,
                           int nkt, const int* __restrict__ pidx, const double* __restrict__ wtpart = nullptr,
                           int nwt = 0, int ldy = 0, int Bt = 0)
{
    // reduces the k-tiles [kt0, kt0 + nkt) of every post tile (the two halves of the two-pass
    // kernels are reduced by separate launches: the first one runs beside pass 2).
    // A block of blockDim.x / 64 waves (1 .. 16) owns one 64-element fragment of G: its chunk partials are one
    // contiguous run of nChunks x 512 bytes (pgl_gpart), the waves take consecutive pieces of it.
    const int P = 1 + DsAll + Nall * B;
    // (Measured round 3 and dropped: splitting the chunks of a fragment over Q blocks whose last arriver -- agent-scope
    //  fence + counter -- combines the partials, to put a 16-fragment reduction on all CUs: the release / acquire fences
    //  (__threadfence in every wave of 800 blocks) write back and invalidate the XCD's L2: +27 us at C1, +140 us at C2.  The same holds for any reduction
    //  "in the tail" of the fused kernels: a kernel boundary is the cheapest cross-XCD synchronisation there is.)
    const int nwf = (int)(blockDim.x >> 6);
    const long long nfrag = (long long)nPT * nkt * 256;
    const int gblocks = (int)((nfrag + 63) / 64);
    __shared__ double red[16][64];
    if ((int)blockIdx.x >= gblocks) {
        // trailing blocks: ll_n and d ll_n / d bias (one block per neuron), when the caller folded the
        // ll reduction into this launch (nsub > 0) -- it then runs beside the G reduction
        if (nsub <= 0) return;
        const int n = (int)blockIdx.x - gblocks;
        if (n >= npost) {
            // ... and behind them (fused stimulus backward): d ll / d w_t[n][bt] = sum over the nwt block partials of
            // k_sepf_finish_d, one block per temporal basis, wave w takes the partials w, w + nwf, ... (fixed order)
            const int bt = n - npost;
            if (wtpart == nullptr || bt >= Bt || grad_out == nullptr) return;
            const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
            double a[4] = {0.0, 0.0, 0.0, 0.0};
            if (lane < ldy) {
                int i = w, u = 0;
                for (; i < nwt; i += nwf, ++u) a[u & 3] += wtpart[((size_t)i * 3 + bt) * ldy + lane];
            }
            red[w][lane] = (a[0] + a[1]) + (a[2] + a[3]);
            __syncthreads();
            if (w == 0 && lane < npost) {
                double v = 0.0;
                for (int j = 0; j < nwf; ++j) v += red[j][lane];
                grad_out[(size_t)lane * P + 1 + bt] = v;
            }
            return;
        }
        pgl_reduce_ll(llpart, gbpart, ll_out, grad_out, n, P, nPT, nChunks, nsub, red);
        return;
    }
    if (grad_out == nullptr) return;
    const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
    const long long gid = blockIdx.x * 64LL + lane;
    const int r = (int)((gid >> 6) & 3);
    const int kt = kt0 + (int)((gid >> 8) % nkt);
    const int pt = (int)((gid >> 8) / nkt);
    const int k = 16 * kt + (lane >> 4) + 4 * r;
    const int n = 16 * pt + (lane & 15);
    const bool live = gid < nfrag && n < npost && k < Ktot;
    double s = 0.0;
    if (live) {
        // wave w sums the chunks [c0, c1) with eight interleaved partial sums (eight loads in flight per
        // lane); the waves' sums are combined in a fixed order: deterministic for a given geometry
        const int per = (nChunks + nwf - 1) / nwf;
        const int c0 = w * per, c1 = (c0 + per < nChunks) ? c0 + per : nChunks;
        const double* gp = Gpart + (((size_t)pt * KT + kt) * 4 + r) * ((size_t)nChunks * 64) + lane;
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += gp[(size_t)(c + j) * 64];
        }
        for (int j = 0; c < c1; ++c, ++j) a[j] += gp[(size_t)c * 64];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && live) {
        s = 0.0;
        for (int j = 0; j < nwf; ++j) s += red[j][lane];
        if (k < Kimp) {
            const int npre = np0 + k / B;
            grad_out[(size_t)n * P + 1 + DsAll + np0 * B + k] = s * Weff[(size_t)npre * Nall + (pidx ? pidx[n] : n_lo + n)];
        } else {
            grad_out[(size_t)n * P + 1 + ds0 + (k - Kimp)] = s;
        }
    }
}

// ll-only evaluations: one block per neuron
__global__ __launch_bounds__(256) void k_finalize_ll(const double* __restrict__ llpart,
                                                     const double* __restrict__ gbpart,
                                                     double* __restrict__ ll_out,
                                                     double* __restrict__ grad_out, int P, int npost,
                                                     int nPT, int nChunks, int nsub)
{
    __shared__ double red[16][64];
    const int n = blockIdx.x;
    if (n >= npost) return;
    pgl_reduce_ll(llpart, gbpart, ll_out, grad_out, n, P, nPT, nChunks, nsub, red);
}

// ---------------------------------------------------------------------------
// Sliced (general) path, phase 2: x = Xbuf + bias -> ll terms and residuals r (in place).
// Thread = one column n of `rows` consecutive bins; per-thread ll / sum(r) partials are reduced
// per neuron by k_rows_reduce (fixed order).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_epilogue(double* __restrict__ Xbuf, int xstride,
                                                       const double* __restrict__ bias,
                                                       const uint8_t* __restrict__ S, int Nall,
                                                       int n_lo, int npost, long long t_lo,
                                                       long long t_hi, int rows, int nlin, double dt,
                                                       double* __restrict__ llp,
                                                       double* __restrict__ gbp,
                                                       const int* __restrict__ pidx)
{
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= npost) return;
    const long long t0 = t_lo + (long long)blockIdx.x * rows;
    long long t1 = t0 + rows;
    if (t1 > t_hi) t1 = t_hi;
    const double b = bias[n];
    double ll = 0.0, gb = 0.0;
    for (long long t = t0; t < t1; ++t) {
        const double x = Xbuf[t * xstride + n] + b;
        const double s = (double)S[t * Nall + (pidx ? pidx[n] : n_lo + n)];
        double term, res;
        pgl_rate_terms(x, s, nlin, dt, term, res, PGL_C);
        ll += term;
        gb += res;
        Xbuf[t * xstride + n] = res;
    }
    llp[(size_t)blockIdx.x * npost + n] = ll;
    gbp[(size_t)blockIdx.x * npost + n] = gb;
}

// rows [t_lo,t_hi) outside the evaluated range must carry r = 0 for the backward launches
__global__ void k_rows_zero(double* __restrict__ Xbuf, int xstride, long long r0, long long r1)
{
    const long long total = (r1 - r0) * xstride;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        Xbuf[r0 * xstride + i] = 0.0;
}

__global__ __launch_bounds__(64) void k_rows_reduce(const double* __restrict__ llp,
                                                    const double* __restrict__ gbp, int nblk,
                                                    int npost, int P, double* __restrict__ ll_out,
                                                    double* __restrict__ grad_out)
{
    const int n = blockIdx.x;
    double a = 0.0, g = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        a += llp[(size_t)b * npost + n];
        g += gbp[(size_t)b * npost + n];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        g += __shfl_xor(g, o, 64);
    }
    if (threadIdx.x == 0) {
        ll_out[n] = a;
        if (grad_out != nullptr) grad_out[(size_t)n * P] = g;
    }
}

// ---------------------------------------------------------------------------
// direct-form helpers (not MFMA): features, impulse currents, state, MCMC inner ll
// ---------------------------------------------------------------------------
// fS[t][n'][b]  (basis.py:201-236).  One block per 16-row tile.
__global__ void k_features(const int2* __restrict__ spk, const int* __restrict__ wlo,
                           const int* __restrict__ whi, const double* __restrict__ phi,
                           double* __restrict__ fS, long long nT, int N, int B, int R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* phiS = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < B * R; i += blockDim.x) phiS[i] = phi[i];
    __syncthreads();
    const int tile = blockIdx.x;
    const int t0 = tile * 16;
    const int K = N * B;
    for (int id = threadIdx.x; id < 16 * K; id += blockDim.x) {
        const int kb = id % K;
        const int t = id / K;
        const long long tg = (long long)t0 + t;
        if (tg >= nT) continue;
        const int np = kb / B, b = kb % B;
        const int lo = wlo[(size_t)tile * N + np];
        const int hi = whi[(size_t)tile * N + np];
        fS[tg * K + kb] = conv_one(spk, lo, hi, (int)tg, R, phiS + b * R);
    }
}

// I_impT[n'][t] = sum_b fS[t,n',b] w[n',b]   (impulse.py:58), transposed for coalescing.
// One block per 64-row tile (4 window tiles of 16).
__global__ void k_impulse_T(const int2* __restrict__ spk, const int* __restrict__ wlo,
                            const int* __restrict__ whi, const double* __restrict__ phi,
                            const double* __restrict__ w, double* __restrict__ IimpT,
                            long long nT, int nT16, int N, int B, int R)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* phiS = reinterpret_cast<double*>(smem);
    double* wS = phiS + B * R;
    for (int i = threadIdx.x; i < B * R; i += blockDim.x) phiS[i] = phi[i];
    for (int i = threadIdx.x; i < N * B; i += blockDim.x) wS[i] = w[i];
    __syncthreads();
    const int t0 = blockIdx.x * 64;
    const int tl0 = blockIdx.x * 4;
    int tl1 = tl0 + 3;
    if (tl1 > nT16 - 1) tl1 = nT16 - 1;
    for (int id = threadIdx.x; id < 64 * N; id += blockDim.x) {
        const int t = id & 63;
        const int np = id >> 6;
        const long long tg = (long long)t0 + t;
        if (tg >= nT) continue;
        const int lo = wlo[(size_t)tl0 * N + np];
        const int hi = whi[(size_t)tl1 * N + np];
        double a = 0.0;
        for (int j = lo; j < hi; ++j) {
            const int2 e = spk[j];
            const int d = (int)tg - e.x - 1;
            if (d >= 0 && d < R) {
                double h = 0.0;
                for (int b = 0; b < B; ++b) h = fma(phiS[b * R + d], wS[np * B + b], h);
                a = fma((double)e.y, h, a);
            }
        }
        IimpT[(size_t)np * nT + tg] = a;
    }
}

// I_net[t] = sum_n' Weff_col[n'] I_impT[n'][t]  (glm.py:39);  I_stim[t] = fstim[t,:].wstim
__global__ void k_inet(const double* __restrict__ IimpT, const double* __restrict__ weff_col,
                       const double* __restrict__ fstim, const double* __restrict__ wstim,
                       double* __restrict__ Inet, double* __restrict__ Istim, long long nT, int N,
                       int Dstim)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nT;
         t += (long long)gridDim.x * blockDim.x) {
        double a = 0.0;
        for (int np = 0; np < N; ++np) a = fma(weff_col[np], IimpT[(size_t)np * nT + t], a);
        Inet[t] = a;
        double s = 0.0;
        for (int j = 0; j < Dstim; ++j) s = fma(fstim[t * Dstim + j], wstim[j], s);
        Istim[t] = s;
    }
}

__global__ void k_lam(const double* __restrict__ Inet, const double* __restrict__ Istim,
                      double bias, int nlin, double* __restrict__ lam, long long nT)
{
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nT;
         t += (long long)gridDim.x * blockDim.x) {
        const double x = bias + Istim[t] + Inet[t];
        double sig, ll;
        lam[t] = (nlin == 1) ? pgl_softplus_parts(x, sig, ll) : exp(x);
    }
}

__global__ void k_axpy(double* __restrict__ y, const double* __restrict__ x, double a, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        y[i] = fma(a, x[i], y[i]);
}

// MCMC inner ll (gibbs.py:910-937): for k < K:
//   x = bias + stim[t] + base[t] - aw_cur*col[t] + w[k]*col[t]
//   ll_k = sum_t -dt*lam_k(t)  +  sum_{spike bins of n_post} S*log(lam_k(t))
// The first sum streams all nT bins (k_ll_current: one exp per element and weight, log1p by
// its series in the |x| > 9.25 tail); the second only visits the post-synaptic neuron's own
// spike events (k_ll_current_spikes, ~2 % of the bins), so no log is evaluated for silent bins.
#define PGL_KMAX 16
__device__ __forceinline__ double pgl_lambda_only(const double x, const int nlin,
                                                  const double* __restrict__ C)
{
    if (nlin != 1) return pgl_exp(x, C);
    const double e = pgl_exp(-fabs(x), C);
    double l1p;
    if (__all(e < C[23])) {                        // |x| > 9.25: alternating series, error < e^6
        l1p = e * fma(-e, fma(-e, fma(-e, fma(-e, 0.2, 0.25), C[22]), 0.5), 1.0);
    } else if (__all(e < 0.1)) {                   // |x| > 2.3: log1p(e) = 2 atanh(s), s = e/(2+e) < 0.048
        const double rc = pgl_rcp(2.0 + e);
        const double s = e * rc;
        const double z = s * s;                    // z < 2.3e-3: z^7/15 < 2.3e-20
        const double q = fma(z, fma(z, fma(z, fma(z, fma(z, fma(z, 1.0 / 13.0, 1.0 / 11.0), 1.0 / 9.0), 1.0 / 7.0),
                                           0.2), C[22]), 1.0);
        l1p = e * ((rc + rc) * q);                 // e last: a denormal e (x ~ -745) must not be halved to zero on the way
    } else {
        const double u = 1.0 + e;
        l1p = pgl_log(u, C) + (e - (u - 1.0)) * pgl_rcp(u);
    }
    return fmax(x, 0.0) + l1p;
}

// candidate weights travel as a kernel argument (<= 16 doubles): no host-to-device copy per batch
struct PglWeights {
    double w[PGL_KMAX];
};

__global__ __launch_bounds__(256) void k_ll_current(const double* __restrict__ base,
                                                    const double* __restrict__ stim,
                                                    const double* __restrict__ colv, double bias,
                                                    double aw_cur, const PglWeights wv,
                                                    int K, int nlin, double dt, long long t_lo,
                                                    long long nT, double* __restrict__ part)
{
    __shared__ double red[4][PGL_KMAX];
    double wk[PGL_KMAX], acc[PGL_KMAX];
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) {
        wk[k] = (k < K) ? wv.w[k] : 0.0;
        acc[k] = 0.0;
    }
    for (long long t = t_lo + blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nT;
         t += (long long)gridDim.x * blockDim.x) {
        const double c = colv[t];
        const double x0 = bias + (stim ? stim[t] : 0.0) + base[t] - aw_cur * c;
#pragma unroll
        for (int k = 0; k < PGL_KMAX; ++k) {
            if (k < K) {
                const double lam = pgl_lambda_only(fma(wk[k], c, x0), nlin, PGL_C);
                // reference semantics: lam == 0 makes log(lam)*S NaN even for S = 0 (glm.py:52)
                acc[k] += (lam == 0.0) ? __builtin_nan("") : -dt * lam;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PGL_KMAX)
        part[(size_t)blockIdx.x * PGL_KMAX + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// spike-bin part: events [e_lo, e_hi) of the post-synaptic neuron, grid-strided; one partial
// row per block (fixed order)
__global__ __launch_bounds__(256) void k_ll_current_spikes(const int2* __restrict__ spk, int e_lo,
                                                           int e_hi, const double* __restrict__ base,
                                                           const double* __restrict__ stim,
                                                           const double* __restrict__ colv,
                                                           double bias, double aw_cur,
                                                           const PglWeights wv, int K,
                                                           int nlin, double* __restrict__ part)
{
    __shared__ double red[4][PGL_KMAX];
    double acc[PGL_KMAX];
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) acc[k] = 0.0;
    for (int i = e_lo + blockIdx.x * blockDim.x + threadIdx.x; i < e_hi; i += gridDim.x * blockDim.x) {
        const int2 e = spk[i];
        const long long t = e.x;
        const double c = colv[t];
        const double x0 = bias + (stim ? stim[t] : 0.0) + base[t] - aw_cur * c;
        const double s = (double)e.y;
        for (int k = 0; k < K; ++k) {
            const double x = fma(wv.w[k], c, x0);
            const double loglam = (nlin == 1) ? pgl_log(pgl_lambda_only(x, nlin, PGL_C), PGL_C) : x;
            acc[k] = fma(s, loglam, acc[k]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PGL_KMAX)
        part[(size_t)blockIdx.x * PGL_KMAX + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[k] = sum over the partial rows; one 64-lane block per k (fixed summation order)
__global__ __launch_bounds__(64) void k_reduce_parts(const double* __restrict__ part, int nblocks,
                                                     int K, double* __restrict__ out)
{
    const int k = blockIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += part[(size_t)b * PGL_KMAX + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[k] = s;
}

// ---------------------------------------------------------------------------
// Collapsed-Gibbs inner ll for MANY post-synaptic columns per launch (gibbs.py:977-1066).  The
// columns (A[:,n], W[:,n]) are conditionally independent given the rest -- the reference maps them
// over its engines (parallel_gibbs.py:162-165) -- so one launch evaluates, for every listed column
// c = (n_post, n_pre), the K candidate weights of the pair:
//   ic[t]   = sum_b fS[t,n_pre,b] * beta[n_post][n_pre][b]        (impulse.py:58 / 308, from the events)
//   x_k[t]  = bias + GX[t][n_post] - aw_cur*ic[t] + w_k*ic[t]     (gibbs.py:914: rank-1 downdate of
//             the resident total current instead of the (nT,N) gemv per pair, gibbs.py:835-864)
//   ll_k    = sum_t -dt*lam_k + S[t,n_post]*log(lam_k)            (gibbs.py:910-937, glm.py:52)
// GX (nT, xs) holds I_stim + I_net of all post neurons (forward-only MFMA pass at prepare time).
// Thread = one column (tid % CP) and every (256/CP)-th bin of the block's rows; four elements per
// lane and pass.  log(lam) is needed for the ~2 % of bins with a spike only: those elements are
// compacted through a per-wave LDS list (rank by ballot), evaluated once per pass by the first
// lanes and read back by their owners -- fixed order, so results are reproducible.
//   part[(bt * ncols + c) * PGL_KMAX + k]
// ---------------------------------------------------------------------------
__device__ __forceinline__ double pgl_pair_current(const int2* __restrict__ spk, const int lo, const int hi,
                                                   const int tg, const int R, const int B,
                                                   const double* __restrict__ phiS,
                                                   const double (&beta)[PGL_MAXB])
{
    double a = 0.0;
    for (int j = lo; j < hi; ++j) {
        const int2 e = spk[j];
        const int d = tg - e.x - 1;
        if (d >= 0 && d < R) {
            double hh = 0.0;
            for (int b = 0; b < B; ++b) hh = fma(phiS[b * R + d], beta[b], hh);
            a = fma((double)e.y, hh, a);
        }
    }
    return a;
}

struct GibbsColsParams {
    const double* __restrict__ GX;       // (nT, xs) I_stim + I_net of every post neuron
    int xs;
    const uint8_t* __restrict__ S;       // (nT, N) counts
    int N, B, R, P, woff;                // woff = 1 + Dstim: first impulse weight of a theta row
    const int2* __restrict__ spk;
    const int* __restrict__ wlo;
    const int* __restrict__ whi;
    const double* __restrict__ phi;      // [B][R]
    const double* __restrict__ theta;    // (N, P) flat feature weights given at prepare time
    const int* __restrict__ cols;        // [ncols] n_post
    const int* __restrict__ pre;         // [ncols] n_pre
    const double* __restrict__ aw;       // [ncols] current A*W of the pair (part of GX)
    const double* __restrict__ w;        // [ncols][K] candidate weights (ll) / [ncols] deltas (update)
    int ncols, CP, K, nlin;
    double dt;
    long long t_lo, t_hi;
    int rows;                            // bins per block
    int gtb;                             // bins per sub-block (multiple of 32, gtb * CP >= 256)
    double* __restrict__ part;
    // regime-split path (k_gibbs_rate_cols + k_gibbs_spike_cols)
    int nsplit;                          // time splits of a block's bins over the waves (narrow launches)
    const int* __restrict__ elo;         // [ncols] events of n_post inside [t_lo, t_hi): first ...
    const int* __restrict__ ehi;         // ... and one past the last index into spk
    double* __restrict__ partS;          // spike-term partials
    int nloop;                           // sub-blocks of PGL_GRB bins per workgroup
    double* __restrict__ hs;             // [ncols][R] impulse response of every listed pair (k_gibbs_cols_setup)
    // launches whose columns all share ONE presynaptic neuron (a sweep step of the collapsed sampler: pair j -> n for
    // every n): its basis-filtered spike train fs[b][t - t_lo] = sum_events count * phi_b[t - s - 1] is built once per
    // launch (k_gibbs_pre_features) and the pair current of a column is B multiply-adds per bin, ic = sum_b fs_b beta_b,
    // instead of a loop over the events of the window per (column, sub-block); null = event loop
    const double* __restrict__ fs;
    long long fs_stride;
    int hs_region;                       // doubles of the first LDS region: max(CP * R, B * (PGL_GRB + 2) + CP * 8)
    int nblkR, nygR, nblkS;              // k_gibbs_rate_cols' 1-D grid: nblkR x nygR rate workgroups, then nblkS x ncols spike workgroups
    int dbg;                             // dev: 1 no event loop, 2 no phase B, 4 no event staging, 8 no GX loads, 16 no band passes,
                                         // 32 no exp, 64 no merge tree (results invalid when != 0)
};

#define PGL_GECAP 24          // staged presynaptic events per column and block (k_gibbs_ll_cols)

// Thread = one (column, candidate weight) pair: one accumulator per thread, a short loop body (the code
// of the first version -- 16 x 4 inlined softplus chains per pass -- did not fit the instruction cache
// and ran at a tenth of the f64 rate).  Per sub-block of PGL_GTB bins the workgroup first builds the
// pair currents of its CP columns in LDS (x0 = bias + I_stim + I_net - aw_cur*ic, ic, spike count; the
// presynaptic events of the block's window are staged in LDS once), then every (column, weight) thread
// walks the bins.  CP = min(ncols, 256 / K) columns per workgroup, grid = (time blocks, column groups).
__global__ __launch_bounds__(256) void k_gibbs_ll_cols(const GibbsColsParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = p.K, CP = p.CP, RPB = 256 / CP;                   // RPB rows per phase-A sweep
    double* phiS = reinterpret_cast<double*>(smem);                 // [B][R]
    double* X0 = phiS + p.B * p.R;                                  // [gtb][CP], gtb * CP >= 256
    double* IC = X0 + p.gtb * CP;
    double* SS = IC + p.gtb * CP;
    int2* evS = reinterpret_cast<int2*>(SS + p.gtb * CP);           // [CP][PGL_GECAP]
    int* ecnt = reinterpret_cast<int*>(evS + (size_t)CP * PGL_GECAP);   // [CP] staged count, -1 = too many
    const int tid = threadIdx.x;
    const long long tb0 = p.t_lo + (long long)blockIdx.x * p.rows;
    long long tb1 = tb0 + p.rows;
    if (tb1 > p.t_hi) tb1 = p.t_hi;
    for (int i = tid; i < p.B * p.R; i += 256) phiS[i] = p.phi[i];
    // presynaptic events that can reach the block's bins: s in [tb0 - R, tb1 - 2], per column
    const int tile_a = (int)(tb0 >> 4), tile_b = (int)((tb1 - 1) >> 4);
    for (int i = tid; i < CP; i += 256) {
        const int cc = blockIdx.y * CP + i;
        int cnt = 0;
        if (cc < p.ncols) {
            const int npc = p.pre[cc];
            cnt = p.whi[(size_t)tile_b * p.N + npc] - p.wlo[(size_t)tile_a * p.N + npc];
        }
        ecnt[i] = (cnt <= PGL_GECAP) ? cnt : -1;
    }
    __syncthreads();
    for (int i = tid; i < CP * PGL_GECAP; i += 256) {
        const int ci = i / PGL_GECAP, j = i % PGL_GECAP;
        const int cc = blockIdx.y * CP + ci;
        if (cc < p.ncols && j < ecnt[ci]) evS[i] = p.spk[p.wlo[(size_t)tile_a * p.N + p.pre[cc]] + j];
    }
    // phase A role: column ca = tid % CP, rows ra, ra + RPB, ... of every sub-block
    const int ca = tid % CP, ra = tid / CP;
    const int cca = blockIdx.y * CP + ca;
    const bool a_valid = (ra < RPB) && (cca < p.ncols);
    const int na = a_valid ? p.cols[cca] : 0, npa = a_valid ? p.pre[cca] : 0;
    const double awa = a_valid ? p.aw[cca] : 0.0;
    const double biasa = a_valid ? p.theta[(size_t)na * p.P] : 0.0;
    double beta[PGL_MAXB];
#pragma unroll
    for (int b = 0; b < PGL_MAXB; ++b)
        beta[b] = (a_valid && b < p.B) ? p.theta[(size_t)na * p.P + p.woff + npa * p.B + b] : 0.0;
    // phase B role: (time split ts, weight kb, column cb), column fastest: the lanes of a wave share the
    // candidate weight (mostly), so the wave-uniform regime tests of the softplus see currents of one
    // sign; TS = 256 / (CP*K) threads share one (column, weight) pair when the launch is narrow (the
    // single-pair evaluations of the ARS draws), each taking every TS-th bin
    const int CK = CP * K;
    const int TS = (256 / CK > 0) ? 256 / CK : 1;
    const int ts_b = tid / CK, q_b = tid % CK;
    const bool worker = ts_b < TS;
    const int kb = worker ? q_b / CP : 0, cb = worker ? q_b % CP : 0;
    const int ccb = blockIdx.y * CP + cb;
    const bool b_valid = worker && ccb < p.ncols;
    const double wk = b_valid ? p.w[(size_t)ccb * K + kb] : 0.0;
    double acc = 0.0;
    __syncthreads();
    const int my_cnt = ecnt[ca];
    const int2* my_ev = evS + (size_t)ca * PGL_GECAP;
    int jlo = 0;                                                    // first staged event still inside the window
    const int GTB = p.gtb;

    for (long long ts = tb0; ts < tb1; ts += GTB) {
        // ---- phase A: pair currents of the sub-block ----
        if (ra < RPB) {
            for (int tt = ra; tt < GTB; tt += RPB) {
                const long long t = ts + tt;
                const bool live = a_valid && t < tb1;
                double a = 0.0, x0 = (p.nlin == 1) ? 30.0 : 0.0, sv = 0.0;   // idle: benign current (series regime)
                if (live) {
                    if (my_cnt >= 0) {
                        // staged events are time-sorted; t grows along the thread's walk, so jlo only moves up
                        for (int j = jlo; j < my_cnt; ++j) {
                            const int2 e = my_ev[j];
                            const int d = (int)t - e.x - 1;
                            if (d < 0) break;
                            if (d >= p.R) {
                                jlo = j + 1;
                                continue;
                            }
                            double hh = 0.0;
                            for (int b = 0; b < p.B; ++b) hh = fma(phiS[b * p.R + d], beta[b], hh);
                            a = fma((double)e.y, hh, a);
                        }
                    } else {
                        const int tile = (int)(t >> 4);
                        a = pgl_pair_current(p.spk, p.wlo[(size_t)tile * p.N + npa], p.whi[(size_t)tile * p.N + npa],
                                             (int)t, p.R, p.B, phiS, beta);
                    }
                    x0 = (biasa + p.GX[t * p.xs + na]) - awa * a;
                    sv = (double)p.S[t * p.N + na];
                }
                X0[tt * CP + ca] = x0;
                IC[tt * CP + ca] = a;
                SS[tt * CP + ca] = sv;
            }
        }
        __syncthreads();
        // ---- phase B: every (time split, weight, column) thread walks its bins ----
        const int nb = (int)((tb1 - ts < GTB) ? tb1 - ts : GTB);
        for (int tt = worker ? ts_b : 0; tt < GTB; tt += TS) {
            const double x = fma(wk, IC[tt * CP + cb], X0[tt * CP + cb]);
            const double sv = SS[tt * CP + cb];
            const double lam = pgl_lambda_only(x, p.nlin, PGL_C);
            // reference semantics: lam == 0 makes log(lam)*S NaN even for S = 0 (glm.py:52)
            double v = (lam == 0.0) ? __builtin_nan("") : -p.dt * lam;
            if (sv != 0.0) v = fma(sv, (p.nlin == 1) ? pgl_log(lam, PGL_C) : x, v);
            acc += (tt < nb) ? v : 0.0;
        }
        __syncthreads();
    }
    // time splits of one (column, weight) pair are summed in split order
    if (TS > 1) {
        double* red = X0;                                           // >= 256 doubles (gtb * CP >= 256)
        red[tid] = acc;
        __syncthreads();
        if (ts_b == 0) {
            acc = 0.0;
            for (int j = 0; j < TS; ++j) acc += red[j * CK + q_b];
        }
    }
    if (b_valid && ts_b == 0) p.part[((size_t)blockIdx.x * p.ncols + ccb) * PGL_KMAX + kb] = acc;
}

// ---------------------------------------------------------------------------
// Regime-split form of the batched inner ll (explinear only).  softplus(x) = max(x,0) + log1p(exp(-|x|)):
//   * |x| >= 12 (the operating regime: bias ~ 20, and the deeply inhibited bins behind a presynaptic spike
//     for the negative quadrature nodes; 87 % of the evaluations at C4): the log1p term is < 6.2e-6 and only
//     needs single precision -- e = v_exp_f32(-|x| log2 e), log1p(e) = e (1 - e/2) (+O(e^3) < 8e-17); its absolute
//     error (<= 1e-6 relative to itself: the f32 rounding of |x| <= 700 in the exponent) is <= 6e-12 of a bin
//     whose rate is >= 12, or of a rate term < 6.2e-6 next to the spike terms.  ~10 instructions per
//     evaluation instead of ~55 of the f64 exp + log1p.
//   * |x| < 12 ("band", 13 % of the evaluations at C4, uniformly distributed in |x|: tools/gibbs_x_hist.py),
//     |x| >= 700 (lam underflows: reference NaN semantics), inf and NaN: the f64 path, on COMPACTED lanes -- band
//     elements are queued per wave in LDS with their weight index and evaluated 64 at a time.
//   * S*log(lam) only exists at the spike bins of n_post (2 % of the bins): k_gibbs_spike_cols walks the
//     event list of the post-synaptic neuron instead of testing every bin.
// Lanes of a wave are 64 consecutive bins of one column; a wave keeps x0 / ic of its bins in registers
// and loops over the K weights (wave-uniform scalar); the band elements of one weight are queued in LDS
// and served before the next weight, the lane partials of a (column, weight) pair are summed by DPP row
// scans (fixed order, no atomics).  ~41 KB of LDS per workgroup: three workgroups per CU.
//   part[blk][c][k] = sum_t lam_k(t)   (k_gibbs_reduce_cols2 applies -dt and adds the spike terms)
// ---------------------------------------------------------------------------
#ifndef PGL_GRB
#define PGL_GRB 256
#endif
//                               bins per sub-block: six segments of 64 per wave and weight -- at C4 13 % of the evaluations are band
                              // elements, ~50 per (item, weight): one f64 pass at ~80 % lane utilisation (four segments: 33 = 52 %)
#ifndef PGL_GFAST
#define PGL_GFAST 12.0f             // (the test reads the single-precision |x|) from |x| >= 12 the log1p(exp(-|x|)) < 6.2e-6 term of the softplus
                                   // comes from the single-precision hardware exp (<= 6e-12 absolute per bin).  Measured round 3:
                                   // switching at |x| >= 8 (three series terms) moves 35 % of the compacted f64
                                   // elements to the fast path -- |x| is uniform below 12 at C4, tools/gibbs_x_hist.py -- but buys
                                   // only 3 % (1.41 vs 1.45 ms) for a 30x larger error (1.7e-10 per bin: the 1e-11 parity test
                                   // against the all-f64 kernel fails), so the threshold stays at 12
#endif
#ifndef PGL_GQ
#define PGL_GQ PGL_GRB
#endif
//                               band-queue entries per wave (the band elements of one weight: <= PGL_GRB)
// Occupancy of k_gibbs_rate_cols (tools/ubench/occ_gibbs_ubench.hip, hipOccupancyMaxActiveBlocksPerMultiprocessor): four
// workgroups per CU up to 40 KB of LDS and 128 VGPRs, three up to 53 KB -- and three are 15-20 % slower.  The kernel sits
// at 40 944 of 40 960 bytes at the C4 shape on purpose (which is why the staged-event cap is 20 and the event counts of
// the windows are 16-bit: the softplus-tail table needs 1.5 KB).  Measured round 3 and dropped because they cross that step or pay more than they
// save: 384-bin sub-blocks (six segments per weight: better filled f64 passes, 46-54 KB) 1.55 ms against 1.40; per-lane
// accumulators for all weights in registers through the VGPR index register (164 VGPRs) 1.53 ms; a band queue whose
// leftovers travel on to the next weights (ring of 320 / 512 entries, three lane-partial vectors in flight, a lane adds
// its result to the weight that owns its queue position: f64 passes at full lanes) 1.44 / 1.71 ms against 1.30 -- the
// bookkeeping around every pass costs more than the half-empty passes it removes.

// log1p(t) for t = exp(-a) in (0, 1] through a 32-interval table (1 KB, copied to LDS): t0 = (2j + 1) / 64,
// {r0 = 1 / (1 + t0), c0 = -t0 r0, L0 = log1p(t0)} rounded from 80-bit values; u = t r0 + c0 = (t - t0) / (1 + t0),
// |u| <= 1/64, log1p(t) = L0 + log1p(u) with eight series terms (u^9 / 9 < 7e-18).  One LDS gather and ten
// multiply-adds instead of frexp + division + 7-term polynomial: absolute error < 2e-16 (1.3e-12 relative at the
// small end, t ~ 6e-6), tools/ubench/proto_math.py.  Only for elements known to lie in the band (a < 12).
__constant__ double PGL_L1PT[32][4] = {
    {0x1.f81f81f81f820p-1, -0x1.f81f81f81f820p-7, 0x1.fc0a8b0fc03e4p-7, 0.0},
    {0x1.e9131abf0b767p-1, -0x1.6ece540f4898dp-5, 0x1.77458f632dcfcp-5, 0.0},
    {0x1.dae6076b981dbp-1, -0x1.28cfc4a33f129p-4, 0x1.341d7961bd1d1p-4, 0.0},
    {0x1.cd85689039b0bp-1, -0x1.93d4bb7e327a9p-4, 0x1.a926d3a4ad563p-4, 0.0},
    {0x1.c0e070381c0e0p-1, -0x1.f8fc7e3f1f8fcp-4, 0x1.0d77e7cd08e59p-3, 0.0},
    {0x1.b4e81b4e81b4fp-1, -0x1.2c5f92c5f92c6p-3, 0x1.44d2b6ccb7d1ep-3, 0.0},
    {0x1.a98ef606a63bep-1, -0x1.59c427e56710ap-3, 0x1.7ab890210d909p-3, 0.0},
    {0x1.9ec8e951033d9p-1, -0x1.84dc5abbf309cp-3, 0x1.af3c94e80bff3p-3, 0.0},
    {0x1.948b0fcd6e9e0p-1, -0x1.add3c0ca4587ep-3, 0x1.e27076e2af2e6p-3, 0.0},
    {0x1.8acb90f6bf3aap-1, -0x1.d4d1bc2503159p-3, 0x1.0a324e27390e3p-2, 0.0},
    {0x1.8181818181818p-1, -0x1.f9f9f9f9f9fa0p-3, 0x1.22941fbcf7966p-2, 0.0},
    {0x1.78a4c8178a4c8p-1, -0x1.0eb66fd0eb670p-2, 0x1.3a64c556945eap-2, 0.0},
    {0x1.702e05c0b8170p-1, -0x1.1fa3f47e8fd20p-2, 0x1.51aad872df82dp-2, 0.0},
    {0x1.6816816816817p-1, -0x1.2fd2fd2fd2fd3p-2, 0x1.686c81e9b14afp-2, 0.0},
    {0x1.6058160581606p-1, -0x1.3f4fd3f4fd3f5p-2, 0x1.7eaf83b82afc3p-2, 0.0},
    {0x1.58ed2308158edp-1, -0x1.4e25b9efd4e26p-2, 0x1.947941c2116fbp-2, 0.0},
    {0x1.51d07eae2f815p-1, -0x1.5c5f02a3a0fd6p-2, 0x1.a9cec9a9a084ap-2, 0.0},
    {0x1.4afd6a052bf5bp-1, -0x1.6a052bf5a814bp-2, 0x1.beb4d9da71b7cp-2, 0.0},
    {0x1.446f86562d9fbp-1, -0x1.7720f353a4c0ap-2, 0x1.d32fe7e00ebd5p-2, 0.0},
    {0x1.3e22cbce4a902p-1, -0x1.83ba68636adfbp-2, 0x1.e744261d68788p-2, 0.0},
    {0x1.3813813813814p-1, -0x1.8fd8fd8fd8fd9p-2, 0x1.faf588f78f31fp-2, 0.0},
    {0x1.323e34a2b10bfp-1, -0x1.9b8396ba9de81p-2, 0x1.0723e5c1cdf40p-1, 0.0},
    {0x1.2c9fb4d812ca0p-1, -0x1.a6c0964fda6c1p-2, 0x1.109f39e2d4c97p-1, 0.0},
    {0x1.27350b8812735p-1, -0x1.b195e8efdb196p-2, 0x1.19ee6b467c96fp-1, 0.0},
    {0x1.21fb78121fb78p-1, -0x1.bc090fdbc0910p-2, 0x1.23130d7bebf43p-1, 0.0},
    {0x1.1cf06ada2811dp-1, -0x1.c61f2a4bafdc6p-2, 0x1.2c0e9ed448e8cp-1, 0.0},
    {0x1.1811811811812p-1, -0x1.cfdcfdcfdcfddp-2, 0x1.34e289d9ce1d3p-1, 0.0},
    {0x1.135c81135c811p-1, -0x1.d946fdd946fdep-2, 0x1.3d9026a7156fbp-1, 0.0},
    {0x1.0ecf56be69c90p-1, -0x1.e26152832c6e0p-2, 0x1.4618bc21c5ec2p-1, 0.0},
    {0x1.0a6810a6810a7p-1, -0x1.eb2fdeb2fdeb3p-2, 0x1.4e7d811b75bb1p-1, 0.0},
    {0x1.0624dd2f1a9fcp-1, -0x1.f3b645a1cac08p-2, 0x1.56bf9d5b3f399p-1, 0.0},
    {0x1.0204081020408p-1, -0x1.fbf7efdfbf7f0p-2, 0x1.5ee02a9241675p-1, 0.0},
};
__device__ __forceinline__ double pgl_log1p_tab(const double t, const double* __restrict__ TB)
{
    const int j = min((int)((float)t * 32.0f), 31);
    const double* te = TB + 4 * j;
    const double u = fma(t, te[0], te[1]);
    const double q = fma(u, fma(u, fma(u, fma(u, fma(u, fma(u, fma(u, -0.125, 1.0 / 7.0), -1.0 / 6.0), 0.2), -0.25),
                                       1.0 / 3.0), -0.5), 1.0);
    return fma(u, q, te[2]);
}

// log1p(exp(-a)) for a in [0, 12] in ONE table step (the band elements of k_gibbs_rate_cols): a0 = j / 8 with
// j = rint(8 a), v = a0 - a (|v| <= 1/16, exact); exp(-a) = E0 (1 + m) with E0 = exp(-a0), m = expm1(v); then
// log1p(E0 (1 + m)) = L0 + log1p(s m) with L0 = log1p(E0), s = E0 / (1 + E0) <= 1/2 from the table ({L0, s} rounded
// from 60-digit values, 97 intervals, 1.5 KB of LDS).  expm1 to v^8 / 8! (next term 4e-17), log1p(w) to w^10 / 10
// (|w| <= 0.0323: next term 4e-18): 26 instructions and one 16-byte LDS gather instead of exp (22) + the log1p table
// (17); absolute error 1.1e-16, relative 2.4e-16 over the whole band (tools/ubench/softplus_tail_table.py) -- the two-step
// form lost relative accuracy at the small end (1.3e-12).
__constant__ double PGL_SPT[97][2] = {
    {0x1.62e42fefa39efp-1, 0x1.0000000000000p-1},
    {0x1.43e4055056374p-1, 0x1.e00aa6681fcf3p-2},
    {0x1.26e18819b6b47p-1, 0x1.c054cda8768f9p-2},
    {0x1.0bd6cffe83c7ap-1, 0x1.a11c01bf10222p-2},
    {0x1.e5746fdb5c064p-2, 0x1.829a0565978dfp-2},
    {0x1.b6fd4f83e1f61p-2, 0x1.65033af8acd79p-2},
    {0x1.8c27e9bc22ee1p-2, 0x1.4885610b9b828p-2},
    {0x1.64cea7ff8a616p-2, 0x1.2d46b08dbbfe4p-2},
    {0x1.40c7abfbec124p-2, 0x1.136561454ba86p-2},
    {0x1.1fe5d241cf50ap-2, 0x1.f5ef21a125693p-3},
    {0x1.01f9b27528a73p-2, 0x1.c81702a88e0d5p-3},
    {0x1.cda525f5dea88p-3, 0x1.9d50402c11d4ap-3},
    {0x1.9c7e908f5420fp-3, 0x1.759b8355a1bb0p-3},
    {0x1.701df494e71dep-3, 0x1.50ee01de5accfp-3},
    {0x1.4823997149a9fp-3, 0x1.2f335e8e7bfd6p-3},
    {0x1.2432d212f7c19p-3, 0x1.104f8e397f508p-3},
    {0x1.03f2d54301d49p-3, 0x1.e84152bac31afp-4},
    {0x1.ce1ebbd958699p-4, 0x1.b501323c9923ap-4},
    {0x1.9a72315646266p-4, 0x1.868d2916eca5bp-4},
    {0x1.6c4bc9f89e092p-4, 0x1.5c90d0f39da16p-4},
    {0x1.4321e1cc6d13fp-4, 0x1.36b7112534847p-4},
    {0x1.1e756ba481cabp-4, 0x1.14abd6d65d0fap-4},
    {0x1.fba37405c85acp-5, 0x1.ec3ad6ad8dc42p-5},
    {0x1.c1984593bfc32p-5, 0x1.b57ae65f9ba04p-5},
    {0x1.8e070fc045701p-5, 0x1.848343c905445p-5},
    {0x1.603f9ae18164ap-5, 0x1.58c85cdebca7bp-5},
    {0x1.37a289e968854p-5, 0x1.31c8280cf1c3dp-5},
    {0x1.13a025a280713p-5, 0x1.0f0a536457387p-5},
    {0x1.e76e4c617c898p-6, 0x1.e040681ccad94p-6},
    {0x1.aee7038d2fdb9p-6, 0x1.a9490c1054030p-6},
    {0x1.7cda8b50a22e0p-6, 0x1.78761313f225ap-6},
    {0x1.508efa245836cp-6, 0x1.4d20122a136cep-6},
    {0x1.295e50b53b654p-6, 0x1.26afa1e43c2c3p-6},
    {0x1.06b48b5ec3195p-6, 0x1.049c3e0cc6678p-6},
    {0x1.d01bb028d8df0p-7, 0x1.ccd6411b606f9p-7},
    {0x1.99e9e19c9117ep-7, 0x1.975c3eecc3be2p-7},
    {0x1.6a033368dd9b7p-7, 0x1.680527a405c3bp-7},
    {0x1.3fae83582545bp-7, 0x1.3e209a7daf6ebp-7},
    {0x1.1a478703e6584p-7, 0x1.191129aaba495p-7},
    {0x1.f27916b786f6ep-8, 0x1.f09503707a24ap-8},
    {0x1.b818da245a728p-8, 0x1.b69f67d638f8ep-8},
    {0x1.84898b1611fd6p-8, 0x1.8363476c064e7p-8},
    {0x1.57008fe54624fp-8, 0x1.561b2d22850c0p-8},
    {0x1.2eca948929bb8p-8, 0x1.2e17c9c24b717p-8},
    {0x1.0b48ec7737a01p-8, 0x1.0abd946147067p-8},
    {0x1.d7de797b8c899p-9, 0x1.d7054b1fc1257p-9},
    {0x1.a082ce8a69e37p-9, 0x1.9fd992191da22p-9},
    {0x1.6fa361566008dp-9, 0x1.6f1f8371cd3fap-9},
    {0x1.447e35674b30ep-9, 0x1.4417772fa800fp-9},
    {0x1.1e67dba01afadp-9, 0x1.1e17cf7f97005p-9},
    {0x1.f991b2f527eb2p-10, 0x1.f914f977dedbfp-10},
    {0x1.be36b6c47edb7p-10, 0x1.bdd58c8bf8274p-10},
    {0x1.89d25404b4136p-10, 0x1.8986a2cac5fa9p-10},
    {0x1.5b93b657d026fp-10, 0x1.5b58bfcb28afcp-10},
    {0x1.32c26f737461cp-10, 0x1.3294815ced7f2p-10},
    {0x1.0ebba110c6b3ap-10, 0x1.0e97da2dda510p-10},
    {0x1.dddef4e20532bp-11, 0x1.dda738adf1189p-11},
    {0x1.a5be000c4a797p-11, 0x1.a592965fa4d74p-11},
    {0x1.743429ab643fap-11, 0x1.7412593d98a3dp-11},
    {0x1.487b7c2fd5f63p-11, 0x1.486125cdb77fcp-11},
    {0x1.21e534d42e269p-11, 0x1.21d0b15711bf3p-11},
    {0x1.ffae1aa2932a2p-12, 0x1.ff8e263314416p-12},
    {0x1.c391acc00d5e5p-12, 0x1.c378c9556743ap-12},
    {0x1.8e84b36ba6fc4p-12, 0x1.8e715105830e4p-12},
    {0x1.5fb2f67077130p-12, 0x1.5fa3dd7d2f7a6p-12},
    {0x1.36612429519aep-12, 0x1.365561fa17242p-12},
    {0x1.11e9e67fdfe4dp-12, 0x1.11e0be0e88435p-12},
    {0x1.e3769e7f0229fp-13, 0x1.e3685aa39565bp-13},
    {0x1.aaa92324cf1c1p-13, 0x1.aa9e06f8cd118p-13},
    {0x1.7887ff5702705p-13, 0x1.787f5839d974fp-13},
    {0x1.4c4a895394428p-13, 0x1.4c43cc2540409p-13},
    {0x1.253fa75ada4a2p-13, 0x1.253a67bfcef61p-13},
    {0x1.02cb0bcfccbe5p-13, 0x1.02c6f5633e446p-13},
    {0x1.c8c588a4de48cp-14, 0x1.c8bf2ab3658e5p-14},
    {0x1.931a243f5bc3cp-14, 0x1.93152ed323578p-14},
    {0x1.63bd0646eb132p-14, 0x1.63b929a27ce33p-14},
    {0x1.39f088538f3eap-14, 0x1.39ed865c5b812p-14},
    {0x1.150d4afbd247ap-14, 0x1.150af353a9a85p-14},
    {0x1.e8ff303b747fbp-15, 0x1.e8fb8a3233ab3p-15},
    {0x1.af8a2784ce55bp-15, 0x1.af8750158434cp-15},
    {0x1.7cd564c9e0d19p-15, 0x1.7cd32e41dd960p-15},
    {0x1.5015d8f26d897p-15, 0x1.50141fba945a1p-15},
    {0x1.28985006982c0p-15, 0x1.2896f8670de67p-15},
    {0x1.05bea3cf0a7cdp-15, 0x1.05bd983178eb2p-15},
    {0x1.cdfa8566527e8p-16, 0x1.cdf8e48f306f6p-16},
    {0x1.97b201e459b7fp-16, 0x1.97b0bd4147285p-16},
    {0x1.67ca56f970021p-16, 0x1.67c95a2556856p-16},
    {0x1.3d83a97525d9ap-16, 0x1.3d82e48dcc901p-16},
    {0x1.1834a8df20647p-16, 0x1.18340f85ba390p-16},
    {0x1.ee8fd2fb90d85p-17, 0x1.ee8ee42004fc9p-17},
    {0x1.b4731c0236f1fp-17, 0x1.b47261fc59032p-17},
    {0x1.812a6eee7f2b6p-17, 0x1.8129de0e79531p-17},
    {0x1.53e86693130f7p-17, 0x1.53e7f5bed75e7p-17},
    {0x1.2bf7bff17c89dp-17, 0x1.2bf76812632a7p-17},
    {0x1.08b88454ae341p-17, 0x1.08b83fe574842p-17},
    {0x1.d33b116773aabp-18, 0x1.d33aa6cf71fcfp-18},
    {0x1.9c5470b033d21p-18, 0x1.9c541dac42246p-18},
};
// (series coefficients by scalar loads: as literals they sit in loop-invariant VGPRs and every Horner step becomes
//  a 64-bit move + v_fmac)
__constant__ double PGL_SPC[16] = {1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0,
                                   -0.1, 1.0 / 9.0, -0.125, 1.0 / 7.0, -1.0 / 6.0, 0.2, -0.25, 1.0 / 3.0, 0.125, 0.0};
template <typename CP>
__device__ __forceinline__ double pgl_softplus_tail_tab(const double a, const double* __restrict__ TB, const CP C)
{
    const double jd = rint(a * 8.0);
    const double v = fma(jd, C[14], -a);
#ifdef PGL_SPT_LINEAR
    // timing ablation only (wrong results): lane-linear, conflict-free table reads instead of the gather
    int zl = 0;
    asm volatile("" : "+v"(zl));
    const double* te = TB + 2 * ((int)(threadIdx.x & 63) + zl);
#else
    const double* te = TB + 2 * (int)jd;
#endif
    double q = fma(v, C[0], C[1]);
#pragma unroll
    for (int i = 2; i <= 5; ++i) q = fma(v, q, C[i]);
    q = fma(v, q, 0.5);
    q = fma(v, q, 1.0);
    const double w = (te[1] * v) * q;
    double pl = fma(w, C[6], C[7]);
#pragma unroll
    for (int i = 8; i <= 13; ++i) pl = fma(w, pl, C[i]);
    pl = fma(w, pl, -0.5);
    pl = fma(w, pl, 1.0);
    return fma(w, pl, te[0]);
}

// h[c][d] = sum_b phi[b][d] * beta[n_post][n_pre][b]: the impulse response of every listed pair, once per launch
__global__ __launch_bounds__(256) void k_gibbs_cols_setup(const GibbsColsParams p)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.ncols * p.R) return;
    const int c = i / p.R, d = i - c * p.R;
    const double* bt = p.theta + (size_t)p.cols[c] * p.P + p.woff + p.pre[c] * p.B;
    double hh = 0.0;
    for (int b = 0; b < p.B; ++b) hh = fma(p.phi[b * p.R + d], bt[b], hh);
    p.hs[i] = hh;
}

#define PGL_GNL 16            // at most this many sub-blocks per workgroup
#define PGL_GECAP_R 20         // staged presynaptic events per column and sub-block (~9 at 20 Hz; more: read from HBM)
#ifdef PGL_SPT_GLOBAL
#define PGL_SPT_N 2
#else
#define PGL_SPT_N 196         // doubles of the softplus-tail table in LDS (97 x 2, padded)
#endif

// Pairwise merges of lane-partial vectors (the reduction tree of k_gibbs_rate_cols).  merge32(a, b): lanes 0..31
// = a[l] + a[l + 32], lanes 32..63 = b[l - 32] + b[l] (v_permlane32_swap: the upper half of the first register
// changes places with the lower half of the second); merge16 the same with rows of 16 lanes (odd rows of the first
// <-> even rows of the second: v_permlane16_swap); merge_dpp<row_mirror, 8> / <row_half_mirror, 4> keep the own
// half / quad of p (lower) resp. q (upper) and add the other one mirrored.
typedef unsigned pgl_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double pgl_merge32(const double a, const double b)
{
    const pgl_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const pgl_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double pgl_merge16(const double a, const double b)
{
    const pgl_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const pgl_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL, int BIT>
__device__ __forceinline__ double pgl_merge_dpp(const double p, const double q, const int lane)
{
    const bool up = (lane & BIT) != 0;
    const double keep = up ? q : p, send = up ? p : q;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(send), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(send), CTRL, 0xf, 0xf, true);
    return keep + __hiloint2double(hi, lo);
}

// sum of v over the 64 lanes of a wave, valid in lane 63: row scans by DPP shifts (zero fill), then the row
// totals travel with row_bcast15 / row_bcast31 -- fixed order, no LDS
__device__ __forceinline__ double pgl_wave_sum_to_last(double v)
{
#define PGL_DPP_ADD(CTRL, ROWMASK)                                                                            \
    {                                                                                                         \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);           \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);           \
        v += __hiloint2double(hi, lo);                                                                        \
    }
    PGL_DPP_ADD(0x111, 0xf)      // row_shr:1
    PGL_DPP_ADD(0x112, 0xf)      // row_shr:2
    PGL_DPP_ADD(0x114, 0xf)      // row_shr:4
    PGL_DPP_ADD(0x118, 0xf)      // row_shr:8   -> lane 15 of every row holds the row total
    PGL_DPP_ADD(0x142, 0xa)      // row_bcast15 -> rows 1 and 3 add the total of the row before
    PGL_DPP_ADD(0x143, 0xc)      // row_bcast31 -> rows 2 and 3 add the total of rows 0-1
#undef PGL_DPP_ADD
    return v;
}

// fs[b][t - t_lo] = sum over the events (s, count) of neuron n_pre of count * phi_b[t - s - 1]  (impulse.py:58: the
// basis-filtered spike train of ONE presynaptic neuron, from its event list); one bin per thread
__global__ __launch_bounds__(256) void k_gibbs_pre_features(const GibbsColsParams p, const int n_pre, double* __restrict__ fs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* phiS = reinterpret_cast<double*>(smem);                 // [B][R]
    for (int i = threadIdx.x; i < p.B * p.R; i += 256) phiS[i] = p.phi[i];
    __syncthreads();
    const long long t = p.t_lo + (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.t_hi) return;
    const long long tile = t >> 4;
    const int lo = p.wlo[(size_t)tile * p.N + n_pre], hi = p.whi[(size_t)tile * p.N + n_pre];
    double acc[PGL_MAXB];
#pragma unroll
    for (int b = 0; b < PGL_MAXB; ++b) acc[b] = 0.0;
    for (int q = lo; q < hi; ++q) {
        const int2 ev = p.spk[q];
        const int d = (int)t - ev.x - 1;
        if ((unsigned)d < (unsigned)p.R) {
            const double cnt = (double)ev.y;
#pragma unroll
            for (int b = 0; b < PGL_MAXB; ++b)
                if (b < p.B) acc[b] = fma(cnt, phiS[b * p.R + d], acc[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < PGL_MAXB; ++b)
        if (b < p.B) fs[(size_t)b * p.fs_stride + (t - p.t_lo)] = acc[b];
}

// softplus(x) - max(x, 0) with the reference's NaN semantics, all-f64: the path of a wave that holds an element near the
// under / overflow of lam.  lam == 0 makes log(lam)*S NaN even for S = 0 (glm.py:52); x >= 700 (incl. +inf): lam = x,
// nothing beyond the max term.  Not inlined: it runs for a handful of waves per launch and would otherwise cost the
// rate loop registers at its 128-VGPR operating point.
__device__ __noinline__ double pgl_gibbs_careful_tail(const double xq)
{
    const double lam = pgl_lambda_only(xq, 1, PGL_C);
    return (xq >= 700.0) ? 0.0 : ((lam == 0.0) ? __builtin_nan("") : lam - __builtin_fmax(xq, 0.0));
}

// (four waves per SIMD = four workgroups per CU is the operating point: the register allocator is held to 128 VGPRs)
__device__ __forceinline__ void pgl_gibbs_rate_body(const GibbsColsParams& p, const int bx, const int by, unsigned char* smem)
{
    constexpr int RB = PGL_GRB, XS = RB + 2, NJ = RB / 32, NSEG = RB / 64;
    const int K = p.K, CP = p.CP, NSPLIT = p.nsplit, RPB = 256 / CP, R = p.R;
    double* HS = reinterpret_cast<double*>(smem);                   // [CP][R] impulse response of the pair
    const bool FSM = p.fs != nullptr;                               // one presynaptic neuron for all columns
    double* FS = HS;                                                // FSM: [B][XS] filtered spike train of the sub-block ...
    double* BT = HS + p.B * XS;                                     // ... and [CP][8] basis weights of the pairs
    double* X0 = HS + p.hs_region;                                  // [CP][XS] bias + I_stim + I_net of the sub-block
    double* Wl = X0 + CP * XS;                                      // [CP][PGL_KMAX]
    double* Qx = Wl + CP * PGL_KMAX;                                // [4][PGL_GQ]
    double* TB = Qx + 4 * PGL_GQ;                                   // [97][2] softplus-tail table (pgl_softplus_tail_tab)
    double* PS = X0;                                                // [CP * NSPLIT][PGL_KMAX] block results: X0 is dead by then
    double* WM = TB + PGL_SPT_N;                                    // [CP] largest |candidate weight| of the column
    int2* evS = reinterpret_cast<int2*>(WM + CP);                   // [CP][PGL_GECAP_R]
    int* WL = reinterpret_cast<int*>(evS + (size_t)CP * PGL_GECAP_R); // [CP][PGL_GNL] first event of the sub-block's window
    unsigned short* WN = reinterpret_cast<unsigned short*>(WL + CP * PGL_GNL);   // [CP][PGL_GNL] events in the window (saturating)
    const int tid = threadIdx.x;
    const long long tw0 = p.t_lo + (long long)bx * RB * p.nloop;
    // ---- once per workgroup: impulse responses, candidate weights, event windows of every sub-block ----
    if (FSM) {
        for (int i = tid; i < CP * 8; i += 256) {
            const int ci = i >> 3, b = i & 7;
            const int cc = by * CP + ci;
            BT[i] = (cc < p.ncols && b < p.B) ? p.theta[(size_t)p.cols[cc] * p.P + p.woff + p.pre[cc] * p.B + b] : 0.0;
        }
    } else {
        for (int i = tid; i < CP * R; i += 256) {
            const int ci = i / R;
            const int cc = by * CP + ci;
            HS[i] = (cc < p.ncols) ? p.hs[(size_t)cc * R + (i - ci * R)] : 0.0;
        }
    }
    for (int i = tid; i < CP * PGL_KMAX; i += 256) {
        const int cc = by * CP + i / PGL_KMAX, k = i % PGL_KMAX;
        Wl[i] = (cc < p.ncols && k < K) ? p.w[(size_t)cc * K + k] : 0.0;
    }
    if (tid < 194 && tid < PGL_SPT_N) TB[tid] = (&PGL_SPT[0][0])[tid];
    for (int i = tid; i < CP * PGL_GNL; i += 256) {
        const int ci = i / PGL_GNL, sb = i % PGL_GNL;
        const int cc = by * CP + ci;
        const long long tb0 = tw0 + (long long)sb * RB;
        int lo = 0, hi = 0;
        if (cc < p.ncols && sb < p.nloop && tb0 < p.t_hi && !FSM) {
            long long tb1 = tb0 + RB;
            if (tb1 > p.t_hi) tb1 = p.t_hi;
            const int npc = p.pre[cc];
            lo = p.wlo[(size_t)(tb0 >> 4) * p.N + npc];
            hi = p.whi[(size_t)((tb1 - 1) >> 4) * p.N + npc];
        }
        WL[i] = lo;
        WN[i] = (unsigned short)((hi - lo < 65535) ? hi - lo : 65535);
    }
    // staging role: column ca, bins ra + j*RPB (j < NJ): eight post neurons of one bin share a 64-byte line
    const int ca = tid % CP, ra = tid / CP;
    const int cca = by * CP + ca;
    const bool a_valid = (ra < RPB) && (cca < p.ncols);
    const int na = a_valid ? p.cols[cca] : 0;
    const double biasa = a_valid ? p.theta[(size_t)na * p.P] : 0.0;
    // evaluation role: a wave owns whole columns; its lanes are consecutive bins
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* const Qxw = Qx + wave * PGL_GQ;
    const int nseg = NSEG / NSPLIT;                                 // 6, 3 or 2 segments of 64 bins per item
    __syncthreads();

    for (int i = tid; i < CP; i += 256) {                           // largest |w_k| of every column (regime pre-check)
        double m = 0.0;
        for (int k = 0; k < K; ++k) m = fmax(m, fabs(Wl[i * PGL_KMAX + k]));
        WM[i] = m;
    }
    __syncthreads();

    // The K lane-partial vectors of an (item, sub-block) -- 64 partial sums each -- are merged pairwise as they
    // appear (pgl_merge32 / 16 / 8 / 4: after four levels ONE vector holds all 16 weights, four lanes each, weight k
    // in the quad bitrev4(k)), added per lane over the workgroup's sub-blocks, and reduced across the quad once per
    // workgroup: 15 merges (57 instructions) per item and sub-block instead of one 24-instruction DPP reduction per
    // weight.  Fixed order, no atomics, no LDS.
    double accV[2] = {0.0, 0.0};
    auto eval_item = [&](const int item, const int sb, const long long tb0, const int nb, double& accv) {
        const int c = item / NSPLIT, sp = item % NSPLIT;
        const int cc = by * CP + c;
        if (cc >= p.ncols || PGL_DBG(2)) return;
        const int tseg = sp * nseg * 64 + lane;                     // bin of segment 0 inside the sub-block
        // pair current of the lane's bins: every event of the column's window adds count * h[t - s - 1]
        double icr[NSEG];
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) icr[sg] = 0.0;
        if (FSM) {
            if (!PGL_DBG(1)) {
                for (int b = 0; b < p.B; ++b) {
                    const double bt = BT[c * 8 + b];                // wave-uniform
#pragma unroll
                    for (int sg = 0; sg < NSEG; ++sg) {
                        const int tt = tseg + 64 * sg;              // (segments beyond the item's share are zeroed below)
                        icr[sg] = fma(FS[b * XS + (tt < RB ? tt : 0)], bt, icr[sg]);
                    }
                }
            }
        } else {
            const int lo = WL[c * PGL_GNL + sb], cnt = WN[c * PGL_GNL + sb];
            const bool staged = cnt <= PGL_GECAP_R;
            const double* hs = HS + c * R;
            const int tr = (int)tb0 + tseg - 1;                     // d = tr + 64*sg - e.x
            if (!PGL_DBG(1)) {
                for (int q = 0; q < cnt; ++q) {
                    const int2 e = staged ? evS[c * PGL_GECAP_R + q] : p.spk[lo + q];
                    const double ecnt = (double)e.y;
#pragma unroll
                    for (int sg = 0; sg < NSEG; ++sg) {
                        const int d = tr + 64 * sg - e.x;
                        if ((unsigned)d < (unsigned)R) icr[sg] = fma(ecnt, hs[d], icr[sg]);
                    }
                }
            }
        }
        const double awc = p.aw[cc];
        const double wmax = WM[c];
        double x0r[NSEG];
        bool bad = false;
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            const int tt = tseg + 64 * sg;
            const bool vl = (sg < nseg) && (tt < nb);
            // lanes without a bin: x = -600 for every weight -- fast regime, exp2f underflows to 0, max(x, 0) = 0:
            // they add exact zeros and need no mask in the weight loop
            x0r[sg] = vl ? X0[c * XS + tt] - awc * icr[sg] : -600.0;
            icr[sg] = vl ? icr[sg] : 0.0;
            // |x_k| <= |x0| + max|w| |ic| for every weight: below 699 no weight reaches the region where lam
            // underflows (reference NaN semantics) or x is inf / NaN
            bad = bad || !(fma(wmax, fabs(icr[sg]), fabs(x0r[sg])) < 699.0);
        }
        // a wave with such a lane sends every element of the item through the f64 path
        const bool careful = __builtin_amdgcn_ballot_w64(bad) != 0ull;
        const unsigned long long all_lanes = __builtin_amdgcn_ballot_w64(true);
        double pend1 = 0.0, pend2 = 0.0, pend3 = 0.0, pend4 = 0.0;
        auto push = [&](const double tot, const int k) {            // k is wave-uniform: scalar branches
            if (!(k & 1)) { pend1 = tot; return; }
            double v = pgl_merge32(pend1, tot);
            if (!(k & 2)) { pend2 = v; return; }
            v = pgl_merge16(pend2, v);
            if (!(k & 4)) { pend3 = v; return; }
            v = pgl_merge_dpp<0x140, 8>(pend3, v, lane);             // row_mirror
            if (!(k & 8)) { pend4 = v; return; }
            accv += pgl_merge_dpp<0x141, 4>(pend4, v, lane);         // row_half_mirror
        };
        double w_next = Wl[c * PGL_KMAX];
        for (int k = 0; k < K; ++k) {
            const double wk = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(w_next)),
                                               __builtin_amdgcn_readfirstlane(__double2loint(w_next)));
            w_next = Wl[c * PGL_KMAX + ((k + 1 < K) ? k + 1 : k)];  // in flight during this iteration
            // softplus(x) = max(x, 0) + log1p(exp(-|x|)): the first term in f64 for every element (one max, one add),
            // the second in single precision where |x| >= PGL_GFAST and through the f64 queue elsewhere.
            // All segments side by side: independent chains.
            double x[NSEG];
            bool bl[NSEG];
            float e[NSEG];
            unsigned long long bm[NSEG], bany = 0ull;               // band lanes of every segment (scalar masks)
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) x[sg] = fma(wk, icr[sg], x0r[sg]);
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                // fast regime read off the single-precision |x| (|x| < 699 is known: no upper bound to test)
                const float af = fabsf((float)x[sg]);
                const bool f = (af >= PGL_GFAST) && !careful;
                bl[sg] = !f;
                // (the ballot of the bare comparison IS its SGPR result; a ballot of bl costs a v_cndmask + v_cmp)
                const unsigned long long mc = __builtin_amdgcn_ballot_w64(!(af >= PGL_GFAST));
                bm[sg] = careful ? all_lanes : mc;
                bany |= bm[sg];
                const float ee = PGL_DBG(32) ? af : __builtin_amdgcn_exp2f(af * -1.44269504088896340736f);
                e[sg] = f ? ee : 0.0f;
            }
            double accl = 0.0;
            float acc1 = 0.0f, acc2 = 0.0f;                         // log1p(e) = e - e^2/2 (+O(e^3) < 8e-17)
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                acc1 += e[sg];
                acc2 = fmaf(e[sg], e[sg], acc2);
                accl += __builtin_fmax(x[sg], 0.0);
            }
            // band elements of this weight: queued per wave, evaluated in f64 on full waves; they contribute
            // lam - max(x, 0) (the max term is already in accl)
            if (bany != 0ull && !PGL_DBG(16)) {
                int qn = 0;
                double accq = 0.0;
#pragma unroll
                for (int sg = 0; sg < NSEG; ++sg) {
                    const unsigned long long m = bm[sg];
                    if (m != 0ull) {
                        const int idx = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (bl[sg]) Qxw[idx] = x[sg];
                        qn += __popcll(m);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                for (int base = 0; base < qn; base += 64) {         // one call site: the f64 code exists once
                    const bool v = base + lane < qn;
                    const double xq = v ? Qxw[base + lane] : 0.0;
                    double d;
                    if (!careful) {
                        // band proper: |x| < 12 is known (single-precision test: <= 12 + 1e-6, table index <= 96)
#ifdef PGL_SPT_GLOBAL
                        d = pgl_softplus_tail_tab(fabs(xq), &PGL_SPT[0][0], (pgl_k_cdp)PGL_SPC);
#else
                        d = pgl_softplus_tail_tab(fabs(xq), TB, (pgl_k_cdp)PGL_SPC);
#endif
                    } else {
                        d = pgl_gibbs_careful_tail(xq);    // rare path, out of line: its registers are not the loop's
                    }
                    accq += v ? d : 0.0;
                }
                __builtin_amdgcn_wave_barrier();
                accl += accq;
            }
            if PGL_DBG(64) accv += accl + (double)fmaf(acc2, -0.5f, acc1); else push(accl + (double)fmaf(acc2, -0.5f, acc1), k);
        }
        for (int k = K; k < PGL_KMAX; ++k) push(0.0, k);            // flush the pending levels
    };

    for (int sb = 0; sb < p.nloop; ++sb) {
        const long long tb0 = tw0 + (long long)sb * RB;
        if (tb0 >= p.t_hi) break;
        long long tb1 = tb0 + RB;
        if (tb1 > p.t_hi) tb1 = p.t_hi;
        const int nb = (int)(tb1 - tb0);
        // ---- staging: the presynaptic events that can reach the sub-block (per column) and
        //      X0 = bias + I_stim + I_net of its bins, [column][bin] ----
        // (requesting the next sub-block's currents into registers before the evaluation and storing them behind it
        //  was measured at 1.63 ms against 1.40: the other two workgroups of the CU already cover this latency)
        if (FSM) {
            for (int i = tid; i < p.B * RB; i += 256) {
                const int b = i / RB, tt = i - b * RB;
                FS[b * XS + tt] = (tt < nb && !PGL_DBG(4)) ? p.fs[(size_t)b * p.fs_stride + (tb0 - p.t_lo) + tt] : 0.0;
            }
        } else {
            for (int i = tid; i < CP * PGL_GECAP_R; i += 256) {
                const int ci = i / PGL_GECAP_R, j = i % PGL_GECAP_R;
                const int lo = WL[ci * PGL_GNL + sb], cnt = WN[ci * PGL_GNL + sb];
                if (cnt <= PGL_GECAP_R && j < cnt && !PGL_DBG(4)) evS[i] = p.spk[lo + j];
            }
        }
        if (ra < RPB) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int tt = ra + j * RPB;
                if (tt < RB)
                    X0[ca * XS + tt] = (a_valid && tt < nb && !PGL_DBG(8)) ? biasa + p.GX[(tb0 + tt) * p.xs + na] : 0.0;
            }
        }
        __syncthreads();
        // ---- evaluation: a wave owns at most two items (column x time split) of the workgroup ----
        for (int half = 0; half < 2; ++half) {                      // rolled: the evaluation code exists once
            const int item = wave + 4 * half;
            if (item >= CP * NSPLIT) break;
            double av = half ? accV[1] : accV[0];
            eval_item(item, sb, tb0, nb, av);
            if (half) accV[1] = av; else accV[0] = av;
        }
        __syncthreads();                                            // X0 / events are rewritten by the next sub-block
    }
    // ---- once per workgroup: the four lanes of every weight's quad (quad_perm butterflies), quad q = bitrev4(k) ----
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int item = wave + 4 * half;
        if (item < CP * NSPLIT) {
            double r = accV[half];
            {
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), 0xB1, 0xf, 0xf, true);
                r += __hiloint2double(hi, lo);
                lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
                hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), 0x4E, 0xf, 0xf, true);
                r += __hiloint2double(hi, lo);
            }
            const int q = lane >> 2;
            const int k = ((q >> 3) & 1) | (((q >> 2) & 1) << 1) | (((q >> 1) & 1) << 2) | ((q & 1) << 3);
            if ((lane & 3) == 0 && k < K) PS[item * PGL_KMAX + k] = r;   // slots owned by this wave
        }
    }
    __syncthreads();
    for (int i = tid; i < CP * PGL_KMAX; i += 256) {
        const int c = i / PGL_KMAX, k = i % PGL_KMAX;
        const int cc = by * CP + c;
        if (cc < p.ncols && k < K) {
            double a = 0.0;
            for (int sp = 0; sp < NSPLIT; ++sp) a += PS[(c * NSPLIT + sp) * PGL_KMAX + k];
            p.part[((size_t)bx * p.ncols + cc) * PGL_KMAX + k] = a;
        }
    }
}

// spike terms of the listed columns: sum over the events (t, count) of n_post inside the evaluated range of
// count * log(lam_k(t)); grid = (event chunks of 256, ncols), one event per thread, f64 throughout.
__device__ __forceinline__ void pgl_gibbs_spike_body(const GibbsColsParams& p, const int bx, const int by, unsigned char* smem)
{
    double (*red)[PGL_KMAX] = reinterpret_cast<double (*)[PGL_KMAX]>(smem);      // [4][PGL_KMAX]
    double* TB = reinterpret_cast<double*>(smem) + 4 * PGL_KMAX;                  // [128] log1p table (pgl_log1p_tab)
    if (threadIdx.x < 128) TB[threadIdx.x] = (&PGL_L1PT[0][0])[threadIdx.x];
    __syncthreads();
    const int tid = threadIdx.x, c = by, K = p.K;
    const int n = p.cols[c], np = p.pre[c];
    const int i = p.elo[c] + bx * 256 + tid;
    double acc[PGL_KMAX];
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) acc[k] = 0.0;
    if (i < p.ehi[c]) {
        const int2 e = p.spk[i];
        const int t = e.x;
        const int tile = t >> 4;
        // pair current at the spike bin from the impulse response of the pair (k_gibbs_cols_setup)
        const double* hs = p.hs + (size_t)c * p.R;
        const int lo = p.wlo[(size_t)tile * p.N + np], hi = p.whi[(size_t)tile * p.N + np];
        double ic = 0.0;
        if (p.fs) {                                                  // shared presynaptic neuron: its filtered spike train
            const double* bt = p.theta + (size_t)n * p.P + p.woff + np * p.B;
            for (int b = 0; b < p.B; ++b) ic = fma(p.fs[(size_t)b * p.fs_stride + (t - p.t_lo)], bt[b], ic);
        } else {
            for (int q = lo; q < hi; ++q) {
                const int2 ev = p.spk[q];
                const int d = t - ev.x - 1;
                if ((unsigned)d < (unsigned)p.R) ic = fma((double)ev.y, hs[d], ic);
            }
        }
        const double x0 = (p.theta[(size_t)n * p.P] + p.GX[(long long)t * p.xs + n]) - p.aw[c] * ic;
        const double sv = (double)e.y;
#pragma unroll
        for (int k = 0; k < PGL_KMAX; ++k) {
            if (k < K) {
                const double x = fma(p.w[(size_t)c * K + k], ic, x0);
                // softplus per lane without wave-uniform regimes (the lanes are unrelated bins): exp, then the table
                // where |x| < 12 and three series terms beyond (e < 6.2e-6: e^4 / 4 is 6e-17 of it); lam = 0 for
                // x < -745 gives log(0) = -inf as the reference expression does
                const double a = fabs(x);
                const double e = pgl_exp(-a, PGL_C);
                const double ser = e * fma(-e, fma(-e, 1.0 / 3.0, 0.5), 1.0);
                const double tab = pgl_log1p_tab(e, TB);
                const double lam = fmax(x, 0.0) + ((a < 12.0) ? tab : ser);
                acc[k] = sv * pgl_log((x != x) ? x : lam, PGL_C);
            }
        }
    }
    // the 16 lane-partial vectors through the pairwise merge tree of k_gibbs_rate_cols (15 merges + a quad butterfly
    // instead of 16 six-step shuffle reductions through the LDS crossbar): weight k ends in quad bitrev4(k)
    const int lane = tid & 63, wave = tid >> 6;
    {
        double m1[8], m2[4], m3[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) m1[i] = pgl_merge32(acc[2 * i], acc[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) m2[i] = pgl_merge16(m1[2 * i], m1[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) m3[i] = pgl_merge_dpp<0x140, 8>(m2[2 * i], m2[2 * i + 1], lane);
        double r = pgl_merge_dpp<0x141, 4>(m3[0], m3[1], lane);
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), 0xB1, 0xf, 0xf, true);
        r += __hiloint2double(hi, lo);
        lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), 0x4E, 0xf, 0xf, true);
        r += __hiloint2double(hi, lo);
        const int q = lane >> 2;
        const int k = ((q >> 3) & 1) | (((q >> 2) & 1) << 1) | (((q >> 1) & 1) << 2) | ((q & 1) << 3);
        if ((lane & 3) == 0) red[wave][k] = r;
    }
    __syncthreads();
    if (tid < K)
        p.partS[((size_t)bx * p.ncols + c) * PGL_KMAX + tid] =
            red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// One launch for both: the rate workgroups (a 1-D grid decoded to the (time block, column group) pairs of the old 2-D grid,
// time block fastest) and BEHIND them the spike workgroups (event chunk, column) -- short, latency-bound chains of dependent
// loads that ran as a launch of their own for 87 us; dispatched last they fill the slots the rate workgroups leave at the
// end of the launch.  (They share nothing but the setup kernels' outputs.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_gibbs_rate_cols(const GibbsColsParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, nrate = p.nblkR * p.nygR;
    if (b < nrate) {
        pgl_gibbs_rate_body(p, b % p.nblkR, b / p.nblkR, smem);
    } else {
        const int bs = b - nrate;
        pgl_gibbs_spike_body(p, bs % p.nblkS, bs / p.nblkS, smem);
    }
}

// out[c][k] = -dt * sum_b part[b][c][k] + sum_b partS[b][c][k] (fixed order); grid = (ncols, K), block = 64
__global__ __launch_bounds__(64) void k_gibbs_reduce_cols2(const double* __restrict__ part, int nblk,
                                                           const double* __restrict__ partS, int nblkS,
                                                           int ncols, int K, double dt, double* __restrict__ out)
{
    const int c = blockIdx.x, k = blockIdx.y;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += part[((size_t)b * ncols + c) * PGL_KMAX + k];
    for (int b = threadIdx.x; b < nblkS; b += 64) q += partS[((size_t)b * ncols + c) * PGL_KMAX + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (threadIdx.x == 0) out[(size_t)c * K + k] = fma(-dt, s, q);
}

// out[c][k] = sum over the time blocks (fixed order); grid = ncols, block = 64
__global__ __launch_bounds__(64) void k_gibbs_reduce_cols(const double* __restrict__ part, int nblk,
                                                          int ncols, int K, double* __restrict__ out)
{
    const int c = blockIdx.x;
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 64) s += part[((size_t)b * ncols + c) * PGL_KMAX + k];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (threadIdx.x == 0) out[(size_t)c * K + k] = s;
    }
}

// GX[t][n_post] += delta_c * ic_c[t] for the listed columns (gibbs.py:1044-1066 writes the new
// (A, W) sample; here the resident total current follows it)
__global__ __launch_bounds__(256) void k_gibbs_update_cols(const GibbsColsParams p, double* __restrict__ GXw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* phiS = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x;
    const int CP = p.CP, RPB = 256 / CP;
    const int cl = tid % CP, rsub = tid / CP;
    const int c = blockIdx.y * CP + cl;
    const bool valid = c < p.ncols;
    for (int i = tid; i < p.B * p.R; i += 256) phiS[i] = p.phi[i];
    __syncthreads();
    if (!valid || rsub >= RPB) return;                   // CP need not divide 256
    const int n = p.cols[c], np = p.pre[c];
    const double delta = p.w[c];
    double beta[PGL_MAXB];
#pragma unroll
    for (int b = 0; b < PGL_MAXB; ++b)
        beta[b] = (b < p.B) ? p.theta[(size_t)n * p.P + p.woff + np * p.B + b] : 0.0;
    const long long tb0 = p.t_lo + (long long)blockIdx.x * p.rows;
    long long tb1 = tb0 + p.rows;
    if (tb1 > p.t_hi) tb1 = p.t_hi;
    for (long long t = tb0 + rsub; t < tb1; t += RPB) {
        const int tile = (int)(t >> 4);
        const double ic = pgl_pair_current(p.spk, p.wlo[(size_t)tile * p.N + np], p.whi[(size_t)tile * p.N + np],
                                           (int)t, p.R, p.B, phiS, beta);
        GXw[t * p.xs + n] = fma(delta, ic, GXw[t * p.xs + n]);
    }
}

// ---------------------------------------------------------------------------
// Stimulus feature build on the device (bkgd.py:122-154, 303-340; basis.py:201-273):
//   1. zx[t,bx] = sum_d interp(stim)[t,d] * basis_x[d,bx]   (np.interp onto the dt grid, then the
//      spatial projection; basis_x == nullptr means identity, Bx == D)
//   2. f[t,bx,bt] = sum_{tau=1..Rt} zx[t-tau,bx] * basis_t[tau-1,bt]   (strictly causal)
// written as fstim[t][col], col = bt*Bx+bx (layout 0, SpatiotemporalStimulus) or bx*Bt+bt
// (layout 1, BasisStimulus: d*B+b).
// ---------------------------------------------------------------------------
__global__ void k_stim_project(const double* __restrict__ stim, long long Tstim, int D,
                               double dt_stim, double dt, const double* __restrict__ basis_x,
                               int Bx, double* __restrict__ zx, long long nT)
{
    const long long total = nT * Bx;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long t = i / Bx;
        const int bx = (int)(i - t * Bx);
        const double x = dt * (double)t;
        // np.interp: clamp outside [xp[0], xp[-1]], else slope form on the bracketing interval
        long long i0 = (long long)floor(x / dt_stim);
        if (i0 > Tstim - 2) i0 = Tstim - 2;
        if (i0 < 0) i0 = 0;
        // guard against x/dt_stim rounding across a knot
        while (i0 + 1 < Tstim - 1 && dt_stim * (double)(i0 + 1) <= x) ++i0;
        while (i0 > 0 && dt_stim * (double)i0 > x) --i0;
        const double x0 = dt_stim * (double)i0, x1 = dt_stim * (double)(i0 + 1);
        const bool beyond = (Tstim < 2) || (x >= dt_stim * (double)(Tstim - 1));
        double acc = 0.0;
        const int d_lo = (basis_x == nullptr) ? bx : 0;          // identity spatial basis: column bx only
        const int d_hi = (basis_x == nullptr) ? bx + 1 : D;
        for (int d = d_lo; d < d_hi; ++d) {
            double v;
            if (beyond) {
                v = stim[(Tstim - 1) * D + d];
            } else {
                const double f0 = stim[i0 * D + d], f1 = stim[(i0 + 1) * D + d];
                v = (f1 - f0) / (x1 - x0) * (x - x0) + f0;
            }
            acc = (basis_x == nullptr) ? v : fma(v, basis_x[(size_t)d * Bx + bx], acc);
        }
        zx[i] = acc;
    }
}

// one block = 256 consecutive bins of one spatial column bx; zx window and basis_t in LDS
__global__ __launch_bounds__(256) void k_stim_conv(const double* __restrict__ zx,
                                                   const double* __restrict__ basis_t, int Rt,
                                                   int Bt, int Bx, int layout,
                                                   double* __restrict__ fstim, long long nT)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* zs = reinterpret_cast<double*>(smem);           // [Rt + 256]
    double* bs = zs + Rt + 256;                             // [Rt][Bt]
    const int bx = blockIdx.y;
    const long long t0 = (long long)blockIdx.x * 256;
    for (int i = threadIdx.x; i < Rt + 256; i += 256) {
        const long long t = t0 - Rt + i;
        zs[i] = (t >= 0 && t < nT) ? zx[t * Bx + bx] : 0.0;
    }
    for (int i = threadIdx.x; i < Rt * Bt; i += 256) bs[i] = basis_t[i];
    __syncthreads();
    const long long t = t0 + threadIdx.x;
    if (t >= nT) return;
    const int Dst = Bx * Bt;
    for (int bt = 0; bt < Bt; ++bt) {
        double a = 0.0;
        // zs index of bin t - tau is threadIdx.x + Rt - tau
        for (int tau = 1; tau <= Rt; ++tau) a = fma(zs[threadIdx.x + Rt - tau], bs[(tau - 1) * Bt + bt], a);
        const int colo = layout == 0 ? bt * Bx + bx : bx * Bt + bt;
        fstim[t * Dst + colo] = a;
    }
}

// ---------------------------------------------------------------------------
// Separable (rank-1) stimulus path for wide stimuli (SpatiotemporalStimulus, bkgd.py:172-345):
//   I_stim[t,n] = sum_{bt,bx} fstim[t,bt,bx] w_t[n,bt] w_x[n,bx]          (bkgd.py:214-227)
// with fstim[t,bt,bx] = sum_tau zx[t-tau,bx] basis_t[tau-1,bt] and zx = interp(stim) . basis_x
// (bkgd.py:303-340, basis.py:238-273).  Interpolation, projection and filtering are linear, so
//   y_n       = interp( (stim . basis_x) . w_x[n] )        a GEMM at the STIMULUS frame rate (T_stim rows)
//   I_stim[:,n] = causal conv of y_n with h_n = basis_t . w_t[n]            (Rt taps)
// and the dense (nT, Bt*Bx) feature matrix (7.4 GB at D_stim = 1024, T = 300 s) is never formed.
// Gradients by the transposes:  rho_n[tau] = sum_t r[t,n] y_n[t-tau]  ->  d/dw_t = basis_t^T rho_n;
//   q_n[s] = sum_tau r[s+tau,n] h_n[tau-1],  Qf = interp^T q_n,  d/dw_x = (stim . basis_x)^T Qf.
// ---------------------------------------------------------------------------
// np.interp of a frame-rate series (clamped at both ends): value at bin t and, for the transpose, the
// bracketing frame and weight.  Same knot logic as k_stim_project.
__device__ __forceinline__ void pgl_interp_knot(const long long t, const double dt, const double dt_stim,
                                                const long long Tstim, long long& i0, double& a, bool& beyond)
{
    const double x = dt * (double)t;
    i0 = (long long)floor(x / dt_stim);
    if (i0 > Tstim - 2) i0 = Tstim - 2;
    if (i0 < 0) i0 = 0;
    while (i0 + 1 < Tstim - 1 && dt_stim * (double)(i0 + 1) <= x) ++i0;
    while (i0 > 0 && dt_stim * (double)i0 > x) --i0;
    beyond = (Tstim < 2) || (x >= dt_stim * (double)(Tstim - 1));
    const double x0 = dt_stim * (double)i0, x1 = dt_stim * (double)(i0 + 1);
    a = (x - x0) / (x1 - x0);
}
__device__ __forceinline__ double pgl_interp_frames(const double* __restrict__ yf, const long long t,
                                                    const double dt, const double dt_stim, const long long Tstim)
{
    if (t < 0) return 0.0;
    long long i0;
    double a;
    bool beyond;
    pgl_interp_knot(t, dt, dt_stim, Tstim, i0, a, beyond);
    if (beyond) return yf[Tstim - 1];
    const double f0 = yf[i0], f1 = yf[i0 + 1];
    return (f1 - f0) * a + f0;
}

// C[m][n] (ldc) = sum_k A[m][k] (lda) * B[n][k] (ldb): small f64 GEMM, 64 x 64 tiles, 256 threads x (4 x 4)
__global__ __launch_bounds__(256) void k_gemm_nt(const double* __restrict__ A, int lda,
                                                 const double* __restrict__ Bm, int ldb,
                                                 double* __restrict__ C, int ldc, int M, int Nn, int Kd)
{
    __shared__ double As[32][65], Bs[32][65];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < Kd; k0 += 32) {
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int r = i >> 5, kk = i & 31;
            As[kk][r] = (m0 + r < M && k0 + kk < Kd) ? A[(size_t)(m0 + r) * lda + k0 + kk] : 0.0;
            Bs[kk][r] = (n0 + r < Nn && k0 + kk < Kd) ? Bm[(size_t)(n0 + r) * ldb + k0 + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[kk][ty * 4 + i];
                b[i] = Bs[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + ty * 4 + i < M && n0 + tx * 4 + j < Nn)
                C[(size_t)(m0 + ty * 4 + i) * ldc + n0 + tx * 4 + j] = acc[i][j];
}

struct SepParams {
    const double* __restrict__ Yf;       // (npost, Tstim) frame-rate projections y_n
    const double* __restrict__ basis_t;  // (Rt, Bt)
    const double* __restrict__ theta;    // (npost, P) rows [bias, w_t(Bt), w_x(Bx), w_imp]
    int P, Bt, Rt, npost, xs;
    long long Tstim, nT, t_lo, t_hi;
    double dt, dt_stim;
    double* __restrict__ X;              // (nT, xs) currents (forward: += I_stim) / residuals r (backward)
};

#define PGL_SEP_TB 1024
// forward: X[t][j] += sum_{tau=1..Rt} y_j[t-tau] h_j[tau-1];  grid = (time blocks, npost), block = 256
__global__ __launch_bounds__(256) void k_sep_conv_fwd(const SepParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ys = reinterpret_cast<double*>(smem);            // [Rt + TB]: bins tb0 - Rt .. tb0 + TB - 1
    double* hs = ys + p.Rt + PGL_SEP_TB;                     // [Rt]
    const int j = blockIdx.y;
    const long long tb0 = p.t_lo + (long long)blockIdx.x * PGL_SEP_TB;
    const double* yf = p.Yf + (size_t)j * p.Tstim;
    for (int i = threadIdx.x; i < p.Rt + PGL_SEP_TB; i += 256)
        ys[i] = pgl_interp_frames(yf, tb0 - p.Rt + i, p.dt, p.dt_stim, p.Tstim);
    for (int i = threadIdx.x; i < p.Rt; i += 256) {
        double h = 0.0;
        for (int b = 0; b < p.Bt; ++b) h = fma(p.basis_t[(size_t)i * p.Bt + b], p.theta[(size_t)j * p.P + 1 + b], h);
        hs[i] = h;
    }
    __syncthreads();
    // thread owns 4 consecutive bins: one new y per tap, four FMAs
    const int o = threadIdx.x * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // out[o+q] = sum_tau ys[Rt + o + q - tau] hs[tau-1]
    double w0 = ys[p.Rt + o - 1 + 0], w1 = ys[p.Rt + o - 1 + 1], w2 = ys[p.Rt + o - 1 + 2], w3 = ys[p.Rt + o - 1 + 3];
    for (int tau = 1; tau <= p.Rt; ++tau) {
        const double h = hs[tau - 1];
        acc[0] = fma(w0, h, acc[0]);
        acc[1] = fma(w1, h, acc[1]);
        acc[2] = fma(w2, h, acc[2]);
        acc[3] = fma(w3, h, acc[3]);
        w3 = w2; w2 = w1; w1 = w0;
        w0 = (tau < p.Rt) ? ys[p.Rt + o - 1 - tau] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long t = tb0 + o + q;
        if (t < p.t_hi && t < p.nT) p.X[t * p.xs + j] += acc[q];
    }
}

// backward, part 1: q_j[s] = sum_{tau=1..Rt} r[s+tau][j] h_j[tau-1]  -> Qb[j][s - t_lo_q]  for the bins s
// that can reach the evaluated range, s in [t_lo - Rt, t_hi);  grid = (time blocks over that range, npost)
__global__ __launch_bounds__(256) void k_sep_conv_bwd(const SepParams p, double* __restrict__ Qb,
                                                      long long s_lo, long long nS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* rs = reinterpret_cast<double*>(smem);            // [TB + Rt]: r of bins sb0 .. sb0 + TB + Rt - 1
    double* hs = rs + p.Rt + PGL_SEP_TB;
    const int j = blockIdx.y;
    const long long sb0 = s_lo + (long long)blockIdx.x * PGL_SEP_TB;
    for (int i = threadIdx.x; i < p.Rt + PGL_SEP_TB; i += 256) {
        const long long t = sb0 + i;
        rs[i] = (t >= p.t_lo && t < p.t_hi) ? p.X[t * p.xs + j] : 0.0;
    }
    for (int i = threadIdx.x; i < p.Rt; i += 256) {
        double h = 0.0;
        for (int b = 0; b < p.Bt; ++b) h = fma(p.basis_t[(size_t)i * p.Bt + b], p.theta[(size_t)j * p.P + 1 + b], h);
        hs[i] = h;
    }
    __syncthreads();
    const int o = threadIdx.x * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // q[o+q'] = sum_tau rs[o + q' + tau] hs[tau-1]
    double w0 = rs[o + 1], w1 = rs[o + 2], w2 = rs[o + 3], w3 = rs[o + 4];
    for (int tau = 1; tau <= p.Rt; ++tau) {
        const double h = hs[tau - 1];
        acc[0] = fma(w0, h, acc[0]);
        acc[1] = fma(w1, h, acc[1]);
        acc[2] = fma(w2, h, acc[2]);
        acc[3] = fma(w3, h, acc[3]);
        w0 = w1; w1 = w2; w2 = w3;
        w3 = (tau < p.Rt) ? rs[o + 4 + tau] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long s = sb0 + o + q;
        if (s - s_lo < nS) Qb[(size_t)j * nS + (s - s_lo)] = (s >= 0) ? acc[q] : 0.0;
    }
}

// backward, part 2: Qf[j][f] = sum_s w(s -> f) q_j[s]  (transpose of np.interp); thread = (f, j), fixed order
__global__ __launch_bounds__(256) void k_sep_interp_T(const SepParams p, const double* __restrict__ Qb,
                                                      long long s_lo, long long nS, double* __restrict__ Qf)
{
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (f >= p.Tstim) return;
    const double ratio = p.dt_stim / p.dt;
    long long a = (long long)floor((double)(f - 1) * ratio) - 2, b = (long long)ceil((double)(f + 1) * ratio) + 2;
    if (f == p.Tstim - 1) b = s_lo + nS;                      // clamped tail: every later bin reads the last frame
    if (a < s_lo) a = s_lo;
    if (a < 0) a = 0;
    if (b > s_lo + nS) b = s_lo + nS;
    double acc = 0.0;
    for (long long s = a; s < b; ++s) {
        long long i0;
        double w;
        bool beyond;
        pgl_interp_knot(s, p.dt, p.dt_stim, p.Tstim, i0, w, beyond);
        double c = 0.0;
        if (beyond) c = (f == p.Tstim - 1) ? 1.0 : 0.0;
        else if (i0 == f) c = 1.0 - w;
        else if (i0 + 1 == f) c = w;
        if (c != 0.0) acc = fma(c, Qb[(size_t)j * nS + (s - s_lo)], acc);
    }
    Qf[(size_t)j * p.Tstim + f] = acc;
}

// backward, part 3: rho_j[tau] = sum_t r[t][j] y_j[t-tau] over one time block -> part[blk][j][tau-1]
// grid = (time blocks over [t_lo, t_hi), npost), block = 256 (thread = lag, looping if Rt > 256)
__global__ __launch_bounds__(256) void k_sep_corr(const SepParams p, double* __restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ys = reinterpret_cast<double*>(smem);            // [Rt + TB]
    double* rs = ys + p.Rt + PGL_SEP_TB;                     // [TB]
    const int j = blockIdx.y;
    const long long tb0 = p.t_lo + (long long)blockIdx.x * PGL_SEP_TB;
    const double* yf = p.Yf + (size_t)j * p.Tstim;
    for (int i = threadIdx.x; i < p.Rt + PGL_SEP_TB; i += 256)
        ys[i] = pgl_interp_frames(yf, tb0 - p.Rt + i, p.dt, p.dt_stim, p.Tstim);
    for (int i = threadIdx.x; i < PGL_SEP_TB; i += 256) {
        const long long t = tb0 + i;
        rs[i] = (t < p.t_hi) ? p.X[t * p.xs + j] : 0.0;
    }
    __syncthreads();
    for (int tau = 1 + threadIdx.x; tau <= p.Rt; tau += 256) {
        double acc = 0.0;
        for (int i = 0; i < PGL_SEP_TB; ++i) acc = fma(rs[i], ys[p.Rt + i - tau], acc);
        part[((size_t)blockIdx.x * p.npost + j) * p.Rt + (tau - 1)] = acc;
    }
}

// d ll / d w_t[j][bt] = sum_tau basis_t[tau-1][bt] * sum_blk part[blk][j][tau-1]; grid = npost, block = 64
__global__ __launch_bounds__(64) void k_sep_wt_grad(const SepParams p, const double* __restrict__ part,
                                                    int nblk, double* __restrict__ grad)
{
    const int j = blockIdx.x;
    for (int b = 0; b < p.Bt; ++b) {
        double s = 0.0;
        for (int tau = threadIdx.x; tau < p.Rt; tau += 64) {
            double rho = 0.0;
            for (int k = 0; k < nblk; ++k) rho += part[((size_t)k * p.npost + j) * p.Rt + tau];
            s = fma(p.basis_t[(size_t)tau * p.Bt + b], rho, s);
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (threadIdx.x == 0) grad[(size_t)j * p.P + 1 + b] = s;
    }
}

// ---------------------------------------------------------------------------
// Separable stimulus at the FRAME rate (dt_stim = q dt, q integer; bkgd.py:303-340 interpolates the stimulus
// linearly between frames, basis.py:238-273 filters it causally).  The interpolated projection y_n is piecewise
// linear over q-bin frames, so the Rt-tap convolution of bin t = q F + o collapses to J = ceil(Rt / q) + 2 frame
// values:
//   I_stim[t,n] = sum_{j<J} sum_{bt} C[row(t)][j][bt] w_t[n,bt] z_n[base(F) + j],   z_n = (stim . basis_x) . w_x[n]
//   base(F) = max(F - M, 0),  M = ceil(Rt / q),  row(t) = t for t < q M (the head, where bins t - tau < 0 are
//   dropped),  q M + o from there on (periodic in the frame);  C is built once per stimulus on the host
//   (build_frame_table): C[row][j][bt] = sum_tau basis_t[tau-1][bt] * (weight of frame base + j in y(t - tau)).
// 15 multiply-adds per bin and neuron at the C5 stress shape instead of 300 taps; the frame index past the last
// frame clamps (np.interp holds the last value).  Currents and residuals travel in the slab layout of the fused
// kernels' accumulators: X[tile - tile0][post tile][r][lane], element (r, lane) = bin 16 tile + (lane >> 4) + 4 r of
// neuron 16 pt + (lane & 15).
//   forward   x[t,n]      = < C[row(t)], ZW_n >,  ZW_n[j][bt] = z_n[base + j] w_t[n,bt]      (k_sepf_fwd)
//   backward  V_n[F][j][bt] = sum_o r[qF+o,n] C[row][j][bt];  d/dw_t[n,bt] = sum_F sum_j V z_n[base + j];
//             H_n[F][j] = sum_bt V w_t[n,bt];  d/dz_n[f] = sum of the H[F][j] with base(F) + j = f   (k_sepf_bwd,
//             k_sepf_finish);  d/dw_x = (stim . basis_x)^T d/dz  (k_gemm_mfma)
// A wave owns one frame F and 64 neurons (4 post tiles): the table row is wave-uniform (scalar loads).
// ---------------------------------------------------------------------------
struct SepfParams {
    const double* __restrict__ Ctab;     // [q (M + 1)][J][BT]
    const double* __restrict__ YfT;      // [Tstim][ldy] frame-rate projections z_n (transposed; written by k_gemm_mfma)
    const double* __restrict__ theta;    // (npost, P) rows [bias, w_t(Bt), w_x(Bx), w_imp]
    double* __restrict__ X;              // slab: currents (forward, written) / residuals (backward, read)
    double* __restrict__ Hb;             // [F1 - F0 + 1][J][ldy]
    double* __restrict__ wpart;          // [workgroups of k_sepf_bwd = ceil(frames / 4)][BT][ldy]
    double* __restrict__ QvT;            // [Tstim][ldy]  d ll / d z_n[f]
    double* __restrict__ grad;           // (npost, P): the w_t columns are written by k_sepf_finish
    int P, Bt, M, q, npost, nPT, ldy, tile0, nTiles;
    long long Tstim, F0, F1;             // frames that hold bins of the tile range
    // fused stimulus backward (k_fused7<.., 3> -> k_sepf_finish_d): pieces D[base - B0][slot][post tile][5][64]
    const double* __restrict__ D;
    long long B0, B1;                    // frame bases of the first / last evaluated tile
    int SL, tilesPerChunk;
};

template <int J, int BT>
__global__ __launch_bounds__(256) void k_sepf_fwd(const SepfParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const int pt = blockIdx.y * 4 + grp;
    const int n = pt * 16 + col;
    const bool vp = pt < p.nPT, vn = vp && n < p.npost;
    const long long F = p.F0 + (long long)blockIdx.x * 4 + wave;
    if (F > p.F1) return;
    const long long base = (F > p.M) ? F - p.M : 0;
    double zw[J][BT];
    {
        double w[BT];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) w[bt] = (vn && bt < p.Bt) ? p.theta[(size_t)n * p.P + 1 + bt] : 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            long long f = base + j;
            if (f > p.Tstim - 1) f = p.Tstim - 1;
            const double z = vn ? p.YfT[(size_t)f * p.ldy + n] : 0.0;
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) zw[j][bt] = z * w[bt];
        }
    }
    const long long tb = (long long)p.tile0 * 16, te = tb + (long long)p.nTiles * 16;
    const long long t0 = F * p.q;
    const int o_lo = (int)((tb > t0) ? tb - t0 : 0), o_hi = (int)((te - t0 < p.q) ? te - t0 : p.q);
    // the table row is wave-uniform: constant address space = scalar loads, the multiply-adds take it from SGPRs
    const pgl_k_cdp crow = (pgl_k_cdp)(p.Ctab + (size_t)(((F < p.M) ? F : p.M) * p.q) * (J * BT));
    double* const xl = p.X + (vp ? (size_t)pt * 256 + col : 0);
#pragma unroll 2
    for (int o = o_lo; o < o_hi; ++o) {
        const pgl_k_cdp cr = crow + (size_t)o * (J * BT);
        double x0 = 0.0, x1 = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                if ((j * BT + bt) & 1)
                    x1 = fma(cr[j * BT + bt], zw[j][bt], x1);
                else
                    x0 = fma(cr[j * BT + bt], zw[j][bt], x0);
            }
        const long long tl = t0 + o - tb;
        if (vp) xl[(size_t)(tl >> 4) * p.nPT * 256 + (size_t)(((tl & 15) >> 2) * 64 + (tl & 3) * 16)] = x0 + x1;
    }
}

template <int J, int BT>
__global__ __launch_bounds__(256) void k_sepf_bwd(const SepfParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const int pt = blockIdx.y * 4 + grp;
    const int n = pt * 16 + col;
    const bool vp = pt < p.nPT, vn = vp && n < p.npost;
    __shared__ double gws[4][BT][64];
    const long long F = p.F0 + (long long)blockIdx.x * 4 + wave;
    const bool vF = F <= p.F1;               // (no early return: the workgroup meets at a barrier below)
    const long long base = (F > p.M) ? F - p.M : 0;
    double V[J][BT];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) V[j][bt] = 0.0;
    const long long tb = (long long)p.tile0 * 16, te = tb + (long long)p.nTiles * 16;
    const long long t0 = F * p.q;
    const int o_lo = (int)((tb > t0) ? tb - t0 : 0), o_hi = vF ? (int)((te - t0 < p.q) ? te - t0 : p.q) : 0;
    const pgl_k_cdp crow = (pgl_k_cdp)(p.Ctab + (size_t)(((F < p.M) ? F : p.M) * p.q) * (J * BT));
    const double* const xl = p.X + (vp ? (size_t)pt * 256 + col : 0);
#pragma unroll 4
    for (int o = o_lo; o < o_hi; ++o) {
        const pgl_k_cdp cr = crow + (size_t)o * (J * BT);
        const long long tl = t0 + o - tb;
        const double r = vn ? xl[(size_t)(tl >> 4) * p.nPT * 256 + (size_t)(((tl & 15) >> 2) * 64 + (tl & 3) * 16)] : 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) V[j][bt] = fma(r, cr[j * BT + bt], V[j][bt]);
    }
    double w[BT], gw[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        w[bt] = (vn && bt < p.Bt) ? p.theta[(size_t)n * p.P + 1 + bt] : 0.0;
        gw[bt] = 0.0;
    }
    const size_t fo = (size_t)(F - p.F0);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        long long f = base + j;
        if (f > p.Tstim - 1) f = p.Tstim - 1;
        const double z = (vn && vF) ? p.YfT[(size_t)f * p.ldy + n] : 0.0;
        double h = 0.0;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            h = fma(V[j][bt], w[bt], h);
            gw[bt] = fma(V[j][bt], z, gw[bt]);
        }
        if (vp && vF) p.Hb[(fo * J + j) * p.ldy + n] = h;
    }
    // d / d w_t: the four frames of the workgroup in a fixed order -> one partial per workgroup
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) gws[wave][bt][lane] = gw[bt];
    __syncthreads();
    if (wave == 0 && vp) {
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
            p.wpart[((size_t)blockIdx.x * BT + bt) * p.ldy + n] =
                ((gws[0][bt][lane] + gws[1][bt][lane]) + gws[2][bt][lane]) + gws[3][bt][lane];
    }
}

// blocks [0, nA): QvT[f][n] = sum of the H[F][j][n] with base(F) + j = f (frames past the last one fold into it),
// 16 frames x 64 neurons per block;  blocks [nA, ...): d ll / d w_t[n][bt] = sum_F wpart[F][bt][n] in a fixed order,
// one block per (64 neurons, bt)
template <int J, int BT>
__global__ __launch_bounds__(1024) void k_sepf_finish(const SepfParams p, const int nA, const int nG)
{
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nF = p.F1 - p.F0 + 1;
    if ((int)blockIdx.x < nA) {
        const int g = blockIdx.x % nG;
        const long long f = (long long)(blockIdx.x / nG) * 16 + wave;
        const int n = g * 64 + lane;
        if (f >= p.Tstim || n >= p.ldy) return;
        auto gather = [&](const long long fv) -> double {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {                 // frames F >= M: base = F - M
                const long long F = fv + p.M - j;
                if (F >= p.M && F >= p.F0 && F <= p.F1) a += p.Hb[((size_t)(F - p.F0) * J + j) * p.ldy + n];
            }
            if (fv < J)                                   // head frames F < M: base = 0, j = fv
                for (long long F = p.F0; F < p.M && F <= p.F1; ++F) a += p.Hb[((size_t)(F - p.F0) * J + fv) * p.ldy + n];
            return a;
        };
        double acc = gather(f);
        if (f == p.Tstim - 1) {
            long long fmax = p.F1 + 1;
            if (fmax < J - 1) fmax = J - 1;
            for (long long fv = f + 1; fv <= fmax; ++fv) acc += gather(fv);
        }
        p.QvT[(size_t)f * p.ldy + n] = acc;
    } else {
        const int b = blockIdx.x - nA;
        const int g = b % nG, bt = b / nG;
        const int n = g * 64 + lane;
        // wave w sums the workgroup partials w, w + 16, ... of k_sepf_bwd (eight independent chains: the loads of a
        // chain are a latency each)
        const long long nW = (nF + 3) / 4;
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (n < p.ldy) {
            long long F = wave;
            for (; F + 16 * 7 < nW; F += 16 * 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += p.wpart[((size_t)(F + 16 * u) * BT + bt) * p.ldy + n];
            }
            for (int u = 0; F < nW; F += 16, ++u) a[u & 7] += p.wpart[((size_t)F * BT + bt) * p.ldy + n];
        }
        red[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
        if (wave == 0 && n < p.npost && bt < p.Bt) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w) v += red[w][lane];
            p.grad[(size_t)n * p.P + 1 + bt] = v;
        }
    }
}

// Fused stimulus backward, second half: k_fused7<.., 3> left D[(j', bt)][n] = sum_i A[i][(j', bt)] r[i][n] per frame base b
// (column (j', bt) belongs to frame min(b + j', Tstim - 1)), one piece per chunk that holds tiles of the base.  Here, per
// frame f and row n (one wave per frame, 16 frames per block, lanes = the <= 64 rows of the four post tiles):
//   Dt[f][bt] = sum of the pieces' columns (f - b, bt) over the bases b = f - 5 .. f (and, for the last frame, of the
//               columns that clamp onto it),
//   QvT[f][n] = sum_bt Dt w_t[n][bt]          (d ll / d z_n[f]: the d/dw_x GEMM follows),
//   wpart[block][bt][n] = sum over the block's frames of Dt z_n[f]   (d ll / d w_t: summed over the blocks by the
//               trailing blocks of k_finalize, in a fixed order).
// Which (base, slot) pieces exist follows from the launch geometry alone (pgl_sepd_first_tile): nothing is zeroed.
__global__ __launch_bounds__(1024) void k_sepf_finish_d(const SepfParams p)
{
    __shared__ double red[16][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane, pt = lane >> 4, col = lane & 15;
    const long long f = (long long)blockIdx.x * 16 + wave;
    const bool vn = n < p.ldy && pt < p.nPT;
    const long long tileE = (long long)p.tile0 + p.nTiles - 1;
    double dt[3] = {0.0, 0.0, 0.0};
    auto gather = [&](const long long fv) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const long long b = fv - j;
            if (b < p.B0 || b > p.B1) continue;
            const long long tf = pgl_sepd_first_tile(b, p.M, p.q, p.tile0);
            long long tl = ((b + 1 + p.M) * (long long)p.q + 15) / 16 - 1;
            if (tl > tileE) tl = tileE;
            const int c0 = (int)((tf - p.tile0) / p.tilesPerChunk), c1 = (int)((tl - p.tile0) / p.tilesPerChunk);
            for (int sl = 0; sl <= c1 - c0; ++sl) {
                const double* dp = p.D + ((((size_t)(b - p.B0) * p.SL + sl) * p.nPT + pt) * 5) * 64 + col;
#pragma unroll
                for (int bt = 0; bt < 3; ++bt) {
                    const int c = 3 * j + bt;
                    const int o = (c < 16) ? (c >> 2) * 64 + (c & 3) * 16 : 4 * 64 + (c - 16) * 16;
                    dt[bt] += dp[o];
                }
            }
        }
    };
    if (vn && f < p.Tstim) {
        gather(f);
        if (f == p.Tstim - 1)
            for (long long fv = f + 1; fv <= p.B1 + 5; ++fv) gather(fv);
    }
    double gw[3] = {0.0, 0.0, 0.0};
    if (vn && f < p.Tstim) {
        const bool vr = n < p.npost;
        const double z = p.YfT[(size_t)f * p.ldy + n];
        double q = 0.0;
#pragma unroll
        for (int bt = 0; bt < 3; ++bt) {
            const double w = (vr && bt < p.Bt) ? p.theta[(size_t)n * p.P + 1 + bt] : 0.0;
            q = fma(dt[bt], w, q);
            gw[bt] = dt[bt] * z;
        }
        p.QvT[(size_t)f * p.ldy + n] = q;
    }
#pragma unroll
    for (int bt = 0; bt < 3; ++bt) red[wave][bt][lane] = gw[bt];
    __syncthreads();
    if (wave < 3 && vn) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w][wave][lane];
        p.wpart[((size_t)blockIdx.x * 3 + wave) * p.ldy + n] = v;
    }
}

// C[m][n] = sum_k A[m sam + k sak] B[n sbn + k sbk] on the f64 MFMA, any strides (8-byte loads); stored at
// C[m scm + n scn].  A workgroup = 8 waves splitting K eight ways in chunks of 32 (lane group kk owns k = 32 u + 8 kk
// + v), each on a 16 (m) x 16 NT (n) tile; the eight partial tiles are added in a fixed order through LDS.  The
// operands of the next chunk are in flight during the MFMAs of the current one (the shapes here -- a few hundred
// tiles, K in the thousands -- are bound by the latency of their load rounds, not by the MFMA rate).
// grid = (ceil(M / 16), ceil(N / (16 NT))).
template <int NT>
__global__ __launch_bounds__(512) void k_gemm_mfma(const double* __restrict__ A, long long sam, long long sak,
                                                   const double* __restrict__ Bm, long long sbn, long long sbk,
                                                   double* __restrict__ C, long long scm, long long scn,
                                                   int M, int Nn, int Kd)
{
    constexpr int NWG = 8, KC = 8;
    __shared__ double red[NWG - 1][NT][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16 * NT;
    const double* ap = A + (size_t)((m0 + i < M) ? m0 + i : M - 1) * sam;
    const double* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = Bm + (size_t)((n0 + 16 * t + i < Nn) ? n0 + 16 * t + i : Nn - 1) * sbn;
    d4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
    const int nU = (Kd + 4 * KC - 1) / (4 * KC);
    double a[KC], b[NT][KC];
    auto fetch = [&](const int u) {
#pragma unroll
        for (int v = 0; v < KC; ++v) {
            const int k = 4 * KC * u + KC * kk + v;
            const bool ok = k < Kd;
            const size_t kc = ok ? k : 0;
            const double av = ap[kc * sak];
            a[v] = ok ? av : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double bv = bp[t][kc * sbk];
                b[t][v] = ok ? bv : 0.0;
            }
        }
    };
    if (wave < nU) fetch(wave);
    for (int u = wave; u < nU; u += NWG) {
        double ca[KC], cb[NT][KC];
#pragma unroll
        for (int v = 0; v < KC; ++v) {
            ca[v] = a[v];
#pragma unroll
            for (int t = 0; t < NT; ++t) cb[t][v] = b[t][v];
        }
        if (u + NWG < nU) fetch(u + NWG);
#pragma unroll
        for (int v = 0; v < KC; ++v)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[v], cb[t][v], acc[t], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][t][r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[t][r];
#pragma unroll
                for (int w = 0; w < NWG - 1; ++w) v += red[w][t][r][lane];
                const int m = m0 + kk + 4 * r, n = n0 + 16 * t + i;      // D[kk + 4 r][i] (see the fused kernels' epilogue)
                if (m < M && n < Nn) C[(size_t)m * scm + (size_t)n * scn] = v;
            }
    }
}

// C[m][n] = sum_k A[m][k] B[n][k] for the two thin GEMMs of the separable stimulus (0.4 GFLOP each, 25 MB streamed once:
// what matters is how the bytes travel, not the MFMA rate).  A is k-contiguous and 16-byte aligned (lda even): a lane
// loads two consecutive k per request (global_load_dwordx4), rows 16 lanes apart, so a wave reads 64-byte runs of 16
// rows and its whole share of K is ONE contiguous run per row (DRAM pages stay open) -- against single 8-byte loads that
// were bound by the texture addresser (k-contiguous) or by DRAM page misses (m-contiguous, rows 24 KB apart).  K even.
// B: BK = 1 k-contiguous and aligned like A (B[n ldb + k]);  BK = 0 n-contiguous (B[k ldb + n], 128-byte lines).
// A workgroup = 8 waves sharing one 16 x 16 NT tile, each wave a contiguous eighth of K in chunks of 8 k (lane group
// kk owns k = 8 c + 2 kk, + 1); NB chunks are requested together (one latency per batch), the eight partial tiles are
// added in a fixed order through LDS.  Stored at C[m scm + n scn].  grid = (ceil(M / 16), ceil(N / (16 NT))).
template <int NT, int BK, int NB>
__global__ __launch_bounds__(512) void k_gemm_kc(const double* __restrict__ A, long long lda,
                                                 const double* __restrict__ Bm, long long ldb,
                                                 double* __restrict__ C, long long scm, long long scn,
                                                 int M, int Nn, int Kd)
{
    constexpr int NWG = 8;
    __shared__ double red[NWG - 1][NT][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16 * NT;
    const double* ap = A + (size_t)((m0 + i < M) ? m0 + i : M - 1) * lda;
    const double* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = (n0 + 16 * t + i < Nn) ? n0 + 16 * t + i : Nn - 1;
        bp[t] = BK ? Bm + (size_t)n * ldb : Bm + n;
    }
    d4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
    const int nC = (Kd + 7) / 8;                          // chunks of 8 k
    const int cpw = (nC + NWG - 1) / NWG;
    const int c_beg = wave * cpw, c_end = (c_beg + cpw < nC) ? c_beg + cpw : nC;
    for (int cb = c_beg; cb < c_end; cb += NB) {
        pgl_d2 a2[NB], b2[NB][NT];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const int k = 8 * (cb + c) + 2 * kk;
            const bool ok = (cb + c < c_end) && k < Kd;   // (Kd is even: a pair is inside or outside as a whole)
            const size_t kc = ok ? k : 0;                  // (a clamped address for the tail)
            const pgl_d2 av = *reinterpret_cast<const pgl_d2*>(ap + kc);
            a2[c].x = ok ? av.x : 0.0;
            a2[c].y = ok ? av.y : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pgl_d2 bv;
                if (BK) {
                    bv = *reinterpret_cast<const pgl_d2*>(bp[t] + kc);
                } else {
                    bv.x = bp[t][kc * ldb];
                    bv.y = bp[t][(kc + 1) * ldb];
                }
                b2[c][t].x = ok ? bv.x : 0.0;
                b2[c][t].y = ok ? bv.y : 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[c].x, b2[c][t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[c].y, b2[c][t].y, acc[t], 0, 0, 0);
            }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][t][r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[t][r];
#pragma unroll
                for (int w = 0; w < NWG - 1; ++w) v += red[w][t][r][lane];
                const int m = m0 + kk + 4 * r, n = n0 + 16 * t + i;
                if (m < M && n < Nn) C[(size_t)m * scm + (size_t)n * scn] = v;
            }
    }
}

// ---------------------------------------------------------------------------
// Spike-triggered average (pyglm/utils/sta.py:6-85) from the event lists:
//   A[i,l,d] = sum_t S[t,n_i] * istim[t-l, d] / sum_t S[t,n_i],  l = 0..L-1 (t-l < 0 -> 0)
// with istim the stimulus interpolated to the bin grid and divided by dt_stim/dt (sta.py:30-41).
// The reference forms a dense (nT, L*D) lag matrix and a gemv per neuron; here only the bins that
// hold spikes are touched.  grid = (ceil(L*D/256), nSel, echunks); block c of the z axis handles an
// equal share of the neuron's events and writes a partial, summed in fixed order by k_sta_finish.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sta(const int2* __restrict__ spk, const int* __restrict__ eoff,
                                             const double* __restrict__ istim, int D, int L,
                                             double* __restrict__ part)
{
    __shared__ int2 ev[256];
    const int i = blockIdx.y, c = blockIdx.z, nSel = gridDim.y, echunks = gridDim.z;
    const long long LD = (long long)L * D;
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(o / D);
    const int d = (int)(o - (long long)l * D);
    const int e0 = eoff[2 * i], e1 = eoff[2 * i + 1];
    const int per = (e1 - e0 + echunks - 1) / echunks;
    const int a = e0 + c * per;
    const int b = min(e1, a + per);
    double acc = 0.0;
    for (int base = a; base < b; base += 256) {
        const int e = base + (int)threadIdx.x;
        ev[threadIdx.x] = (e < b) ? spk[e] : make_int2(0, 0);
        __syncthreads();
        const int m = min(256, b - base);
        if (o < LD) {
            for (int j = 0; j < m; ++j) {
                const int2 q = ev[j];
                const long long tt = (long long)q.x - l;
                if (tt >= 0) acc = fma((double)q.y, istim[tt * D + d], acc);
            }
        }
        __syncthreads();
    }
    if (o < LD) part[((size_t)c * nSel + i) * LD + o] = acc;
}

// A[i][o] = scale * sum_c part[c][i][o] / count_i, count_i = sum of the event counts of neuron i
// (0/0 = NaN for a silent neuron, like the reference's division, sta.py:79-80)
__global__ __launch_bounds__(256) void k_sta_finish(const double* __restrict__ part,
                                                    const int2* __restrict__ spk,
                                                    const int* __restrict__ eoff, long long LD,
                                                    int echunks, double scale, double* __restrict__ A)
{
    __shared__ double red[256];
    const int i = blockIdx.y, nSel = gridDim.y;
    double cnt = 0.0;
    for (int e = eoff[2 * i] + (int)threadIdx.x; e < eoff[2 * i + 1]; e += 256) cnt += (double)spk[e].y;
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    cnt = red[0];
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= LD) return;
    double acc = 0.0;
    for (int c = 0; c < echunks; ++c) acc += part[((size_t)c * nSel + i) * LD + o];
    A[(size_t)i * LD + o] = acc * scale / cnt;
}

// Spike-triggered average at the stimulus FRAME rate (dt_stim = q dt, q integer): np.interp makes the stimulus of bin tt the
// mix (1 - a) stim[f] + a stim[f + 1], f = tt / q, a = (tt % q) / q (the last frame beyond the end), so
//   A[i][l][:] = sum_f Wt[i][l][f] stim[f][:],   Wt[i][l][f] = scale_i * sum over the events (t, c) of neuron i of
//                c ((1 - a) [tt / q == f] + a [tt / q + 1 == f]),  tt = t - l >= 0
// -- a (nSel L) x Tstim weight matrix from the event lists (this kernel: one thread per cell, the few events that can
// reach it found by bisection in the neuron's sorted list) and ONE thin GEMM with the raw stimulus (k_gemm_kc) instead of a
// gather of L x D doubles per spike from the interpolated stimulus: 138 -> 5 ms at 64 neurons x 300 lags x 1024 pixels.
__global__ __launch_bounds__(256) void k_sta_weights(const int2* __restrict__ spk, const int* __restrict__ eoff,
                                                     const double* __restrict__ scale, int L, long long Tstim, int q,
                                                     double* __restrict__ Wt)
{
    const int i = blockIdx.z, l = blockIdx.y;
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= Tstim) return;
    const int e0 = eoff[2 * i], e1 = eoff[2 * i + 1];
    // events with tt = t - l in [(f - 1) q, (f + 1) q); the last frame also takes everything beyond it
    const long long lo_t = (f - 1) * (long long)q + l, hi_t = (f + 1 >= Tstim) ? (1ll << 62) : (f + 1) * (long long)q + l;
    int a = e0, b = e1;
    while (a < b) {                                       // first event with t >= lo_t (and t >= l)
        const int m = (a + b) >> 1;
        if ((long long)spk[m].x < lo_t) a = m + 1; else b = m;
    }
    double acc = 0.0;
    for (int e = a; e < e1; ++e) {
        const int2 ev = spk[e];
        if ((long long)ev.x >= hi_t) break;
        const long long tt = (long long)ev.x - l;
        if (tt < 0) continue;
        const long long fr = tt / q;
        const double al = (double)(tt - fr * q) / (double)q;
        double w;
        if (fr >= Tstim - 1) w = (f == Tstim - 1) ? 1.0 : 0.0;       // np.interp holds the last frame
        else w = (fr == f) ? 1.0 - al : ((fr + 1 == f) ? al : 0.0);
        acc = fma((double)ev.y, w, acc);
    }
    Wt[((size_t)i * L + l) * Tstim + f] = acc * scale[i];
}

// transpose of the uint8 count matrix: ST[n][t] = S[t][n]
__global__ void k_transpose_u8(const uint8_t* __restrict__ S, uint8_t* __restrict__ ST,
                               long long nT, int N)
{
    __shared__ uint8_t tile[64][65];
    const long long t0 = (long long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int tt = i / 64, nn = i % 64;
        const long long t = t0 + tt;
        const int n = n0 + nn;
        tile[tt][nn] = (t < nT && n < N) ? S[t * N + n] : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int nn = i / 64, tt = i % 64;
        const long long t = t0 + tt;
        const int n = n0 + nn;
        if (t < nT && n < N) ST[(size_t)n * nT + t] = tile[tt][nn];
    }
}

// ---------------------------------------------------------------------------
// Lock-step BFGS (inference/batched_bfgs.py): the per-neuron optimiser state machines of all M neurons of a shard as a
// handful of row kernels -- one workgroup per neuron row -- around the fused ll+grad launch.  The reference calls
// scipy.optimize.minimize(method="bfgs") per neuron (coord_descent.py:194-199); the row kernels run the same algorithm
// for every neuron at once: BFGS from H = I, More'-Thuente line search for the strong Wolfe conditions
// (pglm_linesearch.h: scipy's DCSRCH with scipy's constants and first trial step), termination on max|g| <= gtol or
// maxiter iterations.  Where scipy gives up ("precision loss": the search reports a warning) the row takes the best
// sufficient-decrease step of that search if there is one, else restarts once from steepest descent, then freezes.
// All state lives in ONE device block of doubles (flags and counters included), laid out by pgl_bfgs_view; the dense
// inverse Hessians H (M, P, ld) are the caller's buffer and are touched by k_bfgs_hmul only: ONE read-modify-write
// pass per accepted iteration applies the pending rank-3 update H += U V^T of the previous iteration and multiplies
// by the new gradient; a (re)started H = hscale * I is never materialised before its first update.
// ---------------------------------------------------------------------------
#include "pglm_linesearch.h"

struct BfgsView {
    int M, P;
    double *X, *g, *p, *Hg, *s, *y, *t, *Xb, *gb;    // (M, P)
    double *U, *V;                                   // (M, P, 3): H += U V^T is the BFGS update
    double *f, *fprev, *alpha, *slope, *rho, *hscale, *iters, *restarts, *active, *frozen, *acc, *upd, *stall, *ident,
           *pend, *fb, *nfev, *hk;                   // (M)
    double* ls;                                      // (PGL_LS_NDOUBLES, M): line-search state, field-major
};
#define PGL_BFGS_NVEC 9
#define PGL_BFGS_NSCAL 18
#define PGL_LS_FTOL 1e-4
#define PGL_LS_GTOL 0.9
#define PGL_LS_XTOL 1e-14
#define PGL_LS_STPMIN 1e-100
#define PGL_LS_STPMAX 1e100
__host__ __device__ inline size_t pgl_bfgs_doubles(int M, int P)
{
    return (size_t)M * P * (PGL_BFGS_NVEC + 6) + (size_t)M * (PGL_BFGS_NSCAL + PGL_LS_NDOUBLES);
}
__host__ __device__ inline BfgsView pgl_bfgs_view(double* st, int M, int P)
{
    BfgsView v;
    const size_t MP = (size_t)M * P;
    v.M = M; v.P = P;
    v.X = st; v.g = st + MP; v.p = st + 2 * MP; v.Hg = st + 3 * MP; v.s = st + 4 * MP; v.y = st + 5 * MP;
    v.t = st + 6 * MP; v.Xb = st + 7 * MP; v.gb = st + 8 * MP; v.U = st + 9 * MP; v.V = st + 12 * MP;
    double* q = st + 15 * MP;
    v.f = q; v.fprev = q + M; v.alpha = q + 2 * M; v.slope = q + 3 * M; v.rho = q + 4 * M; v.hscale = q + 5 * M;
    v.iters = q + 6 * M; v.restarts = q + 7 * M; v.active = q + 8 * M; v.frozen = q + 9 * M; v.acc = q + 10 * M;
    v.upd = q + 11 * M; v.stall = q + 12 * M; v.ident = q + 13 * M; v.pend = q + 14 * M; v.fb = q + 15 * M;
    v.nfev = q + 16 * M; v.hk = q + 17 * M;
    v.ls = q + (size_t)PGL_BFGS_NSCAL * M;
    return v;
}
#define PGL_LS_FIELDS(F) F(stp, 0) F(finit, 1) F(ginit, 2) F(gtest, 3) F(stx, 4) F(fx, 5) F(gx, 6) F(sty, 7) F(fy, 8) \
    F(gy, 9) F(stmin, 10) F(stmax, 11) F(width, 12) F(width1, 13) F(brackt, 14) F(stage, 15) F(nfev, 16) F(moved, 17)
__device__ __forceinline__ void pgl_ls_load(const BfgsView& v, int r, PglLs* s)
{
#define PGL_LS_LD(name, k) s->name = v.ls[(size_t)k * v.M + r];
    PGL_LS_FIELDS(PGL_LS_LD)
#undef PGL_LS_LD
}
__device__ __forceinline__ void pgl_ls_store(const BfgsView& v, int r, const PglLs* s)
{
#define PGL_LS_ST(name, k) v.ls[(size_t)k * v.M + r] = s->name;
    PGL_LS_FIELDS(PGL_LS_ST)
#undef PGL_LS_ST
}

// sum / max over the 256 threads of a block, result in every thread (fixed order)
// (blocks of more than 256 threads -- the one-launch iteration k_bfgs_step<1024> -- run their row loops on the first 256
//  threads only, so a row's sums are the same numbers whichever kernel computes them; the other threads pass through the
//  barriers with nothing to add)
__device__ __forceinline__ double pgl_blk_sum(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double pgl_blk_max(double v, double* red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// first column of a thread's stride-256 walk over a row of P numbers (threads beyond the first 256: none)
__device__ __forceinline__ int pgl_row_c0(const int tid, const int P) { return tid < 256 ? tid : P; }

// start of a fit: X, f, g of every row are in place; H = I (not materialised), steepest-descent direction, scipy's
// first trial step min(1, 1.01 / |g|) (its old_old_fval = f + |g| / 2), rows with max|g| <= gtol never start
__global__ __launch_bounds__(256) void k_bfgs_init(const BfgsView v, const double gtol)
{
    __shared__ double red[4];
    const int r = blockIdx.x, tid = threadIdx.x, P = v.P;
    const size_t o = (size_t)r * P;
    double gg = 0.0, gmax = 0.0;
    for (int c = tid; c < P; c += 256) {
        const double gc = v.g[o + c];
        v.Hg[o + c] = gc;
        v.p[o + c] = -gc;
        gg = fma(gc, gc, gg);
        gmax = fmax(gmax, fabs(gc));
    }
    gg = pgl_blk_sum(gg, red);
    gmax = pgl_blk_max(gmax, red);
    if (tid == 0) {
        const double f = v.f[r], fprev = f + sqrt(gg) / 2.0, slope = -gg;
        v.fprev[r] = fprev; v.slope[r] = slope; v.rho[r] = 0.0; v.hscale[r] = 1.0; v.iters[r] = 0.0; v.restarts[r] = 0.0;
        v.frozen[r] = 0.0; v.acc[r] = 0.0; v.upd[r] = 0.0; v.stall[r] = 0.0; v.ident[r] = 1.0; v.pend[r] = 0.0;
        v.fb[r] = f; v.nfev[r] = 0.0; v.hk[r] = 0.0;
        v.active[r] = gmax > gtol ? 1.0 : 0.0;
        PglLs s;
        const double a0 = pgl_ls_first_step(f, fprev, slope);
        pgl_ls_start(&s, a0, f, slope, PGL_LS_FTOL, PGL_LS_STPMIN, PGL_LS_STPMAX);
        pgl_ls_store(v, r, &s);
        v.alpha[r] = a0;
    }
}

// trial points of the listed rows: Xt[j] = X[r] + alpha[r] p[r], r = rows[j] (null: r = j)
__global__ __launch_bounds__(256) void k_bfgs_trial(const BfgsView v, const int* __restrict__ rows,
                                                    double* __restrict__ Xt)
{
    const int j = blockIdx.x, r = rows ? rows[j] : j;
    const double a = v.alpha[r];
    for (int c = threadIdx.x; c < v.P; c += 256)
        Xt[(size_t)j * v.P + c] = fma(a, v.p[(size_t)r * v.P + c], v.X[(size_t)r * v.P + c]);
}

struct BfgsPrior {
    int N, B, Dstim, kind;                // kind: 0 Gaussian, 1 group lasso on the impulse weights (priors.py:139 / 202)
    double mu_b, sg_b, stim_sigma, mu, sigma, lam;
};

// f = -(ll + log prior), g = -(grad ll + grad log prior) of one trial row x = [bias, w_stim, w_ir] (the packing that IS the
// device's theta row); fit_glm's NaN rules: f NaN -> 1e16, any NaN in a gradient row -> zero row
// (coord_descent.py:170-182).  In place: *llj -> f, g (grad) -> g.  Whole block; row loops on its first 256 threads.
__device__ __forceinline__ void pgl_bfgs_objective_row(const int P, const double* __restrict__ x, double* __restrict__ g,
                                                       double* __restrict__ llj, const BfgsPrior& q, double* red, const int tid)
{
    double lp = 0.0;
    bool bad = false;
    if (tid == 0) {                                                            // bias.py:33
        const double d = x[0] - q.mu_b;
        lp += -0.5 / (q.sg_b * q.sg_b) * d * d;
        const double gv = -(g[0] - d / (q.sg_b * q.sg_b));
        bad = bad || (gv != gv);
        g[0] = gv;
    }
    for (int c = 1 + pgl_row_c0(tid, q.Dstim); c < 1 + q.Dstim; c += 256) {    // bkgd.py:76
        const double w = x[c], is2 = 1.0 / (q.stim_sigma * q.stim_sigma);
        lp += -0.5 * is2 * w * w;
        const double gv = -(g[c] - w * is2);
        bad = bad || (gv != gv);
        g[c] = gv;
    }
    const int o = 1 + q.Dstim;
    for (int n = pgl_row_c0(tid, q.N); n < q.N; n += 256) {                    // one presynaptic group per thread
        const double* w = x + o + n * q.B;
        double* gw = g + o + n * q.B;
        if (q.kind == 1) {                                                     // priors.py:202
            double z[PGL_MAXB], ss = 0.0;
            for (int b = 0; b < q.B; ++b) {
                z[b] = (w[b] - q.mu) / q.sigma;
                ss += z[b] * z[b];
            }
            const double nrm = sqrt(ss);
            lp -= q.lam * nrm;
            for (int b = 0; b < q.B; ++b) {
                const double gv = -(gw[b] - q.lam * z[b] / nrm / q.sigma);     // 0/0 -> NaN like the host prior
                bad = bad || (gv != gv);
                gw[b] = gv;
            }
        } else {                                                               // priors.py:139
            const double is2 = 1.0 / (q.sigma * q.sigma);
            for (int b = 0; b < q.B; ++b) {
                const double d = w[b] - q.mu;
                lp += -0.5 * is2 * d * d;
                const double gv = -(gw[b] - d * is2);
                bad = bad || (gv != gv);
                gw[b] = gv;
            }
        }
    }
    const double lpt = pgl_blk_sum(lp, red);
    const bool anybad = pgl_blk_max(bad ? 1.0 : 0.0, red) > 0.0;
    if (tid == 0) {
        const double fv = -(*llj + lpt);
        *llj = (fv != fv) ? 1e16 : fv;
    }
    if (anybad)
        for (int c = pgl_row_c0(tid, P); c < P; c += 256) g[c] = 0.0;
}

// (pgl_bfgs_objective_dev: the objective of rows that are not in a search -- the starting point of a fit)
__global__ __launch_bounds__(256) void k_bfgs_objective(const int P, const double* __restrict__ Xt,
                                                        double* __restrict__ ll, double* __restrict__ grad,
                                                        const BfgsPrior q)
{
    __shared__ double red[4];
    const int j = blockIdx.x;
    pgl_bfgs_objective_row(P, Xt + (size_t)j * P, grad + (size_t)j * P, ll + j, q, red, (int)threadIdx.x);
}

// One line-search step of a row whose search is running (r: its row of the state, xt / ftj / gt: the evaluated trial):
// phi'(alpha) = g_trial . p, then the More'-Thuente state machine.  Outcomes: another trial step (alpha[r]); the trial
// satisfies the strong Wolfe conditions and the row takes it (X, f, g; s, y, rho left behind, acc = 1); or the search cannot
// make progress -- scipy stops there ("precision loss"; identical iterates up to that first warning only) -- then the best
// sufficient-decrease point of this search is taken if there is one (the trial itself, or the best step saved so far, Xb / gb
// -- judged by ITS step length, also when the search is cut off by max_trials on a call in which the trial became the best
// step), else stall = 1 (the update phase restarts or freezes the row).
__device__ __forceinline__ void pgl_bfgs_linesearch_row(const BfgsView& v, const int r, const double* __restrict__ xt,
                                                        const double* __restrict__ ftj, const double* __restrict__ gt,
                                                        const int max_trials, double* red, int* dec, const int tid)
{
    const int P = v.P;
    if (v.active[r] == 0.0) return;
    const size_t o = (size_t)r * P;
    double dp = 0.0;
    for (int c = pgl_row_c0(tid, P); c < P; c += 256) dp = fma(gt[c], v.p[o + c], dp);
    dp = pgl_blk_sum(dp, red);
    if (tid == 0) {
        // (the state machine is wave-uniform scalar code; routed through a vector register index so that its ~40 doubles
        // live in VGPRs instead of spilling the scalar register file)
        int rv = r;
        asm volatile("" : "+v"(rv));
        const int r = rv;
        PglLs s;
        pgl_ls_load(v, r, &s);
        const double stp = s.stp, stx_prev = s.stx;
        const double f = *ftj;
        // an infinite objective or slope ends the search like scipy's ("WARN": its fallback search fails on inf as well);
        // NaN never arrives here (fit_glm's rule: 1e16 and a zero gradient, applied by the objective)
        int rc = PGL_LS_WARNING;
        if (f - f == 0.0 && dp - dp == 0.0)
            rc = pgl_ls_step(&s, f, dp, PGL_LS_FTOL, PGL_LS_GTOL, PGL_LS_XTOL, PGL_LS_STPMIN, PGL_LS_STPMAX);
        else s.moved = 0.0;
        if (rc == PGL_LS_EVALUATE && s.nfev >= (double)max_trials) rc = PGL_LS_WARNING;
        int src = 0;                                         // 1: take the trial, 2: take the saved best step
        const int moved = s.moved != 0.0;
        if (rc == PGL_LS_CONVERGED) src = 1;
        else if (rc == PGL_LS_WARNING) {
            // (Xb / gb / fb still hold the best step BEFORE this call -- they are only overwritten on EVALUATE -- and
            //  stx_prev is its step length: s.stx is the trial's when it has just become the best step)
            const bool okT = stp > 0.0 && f <= s.finit + stp * s.gtest && f < s.finit;
            const bool okB = stx_prev > 0.0 && v.fb[r] <= s.finit + stx_prev * s.gtest && v.fb[r] < s.finit;
            if (okT && (!okB || f <= v.fb[r])) src = 1;
            else if (okB) src = 2;
            if (src == 0) v.stall[r] = 1.0;
        } else {
            pgl_ls_store(v, r, &s);
            v.alpha[r] = s.stp;
            if (moved) v.fb[r] = f;
        }
        v.nfev[r] += 1.0;
        dec[0] = rc; dec[1] = src; dec[2] = moved;
    }
    __syncthreads();
    const int rc = dec[0], src = dec[1];
    if (rc == PGL_LS_EVALUATE) {
        if (dec[2])                                          // the trial is the best step so far: keep its point and gradient
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
                v.Xb[o + c] = xt[c];
                v.gb[o + c] = gt[c];
            }
        return;
    }
    if (src == 0) return;
    const double* xs = src == 1 ? xt : v.Xb + o;
    const double* gs = src == 1 ? gt : v.gb + o;
    double sy = 0.0;
    for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
        const double xn = xs[c], gn = gs[c];
        const double s = xn - v.X[o + c], y = gn - v.g[o + c];
        v.s[o + c] = s;
        v.y[o + c] = y;
        v.X[o + c] = xn;
        v.g[o + c] = gn;
        sy = fma(s, y, sy);
    }
    sy = pgl_blk_sum(sy, red);
    if (tid == 0) {
        const double fn = src == 1 ? *ftj : v.fb[r];
        v.fprev[r] = v.f[r];
        v.f[r] = fn;
        v.acc[r] = 1.0;
        const double rho = 1.0 / sy;
        const bool u = sy > 0.0 && rho - rho == 0.0;         // curvature condition holds (always after a Wolfe step)
        v.upd[r] = u ? 1.0 : 0.0;
        v.rho[r] = u ? rho : 0.0;
    }
}

// t = H g for the listed rows that have just taken a step (acc = 1), in the same pass over H that applies the pending
// rank-3 update of the previous iteration: H <- H + U V^T (pend), t = H g.  A row whose H is still hscale * I (ident)
// is materialised here together with its first update; without a pending update it is not touched at all
// (k_bfgs_update uses t = hscale * g).  Grid (ceil(P / 32), L); a wave owns 8 rows of H, lanes run along the columns
// in 16-byte pieces (ld even).  Traffic: one read + one write of P x ld doubles per row and accepted iteration.
#define PGL_HM_ROWS 8
__global__ __launch_bounds__(256) void k_bfgs_hmul(const BfgsView v, const int* __restrict__ rows,
                                                   double* __restrict__ H, const int ld)
{
    const int j = blockIdx.y, r = rows ? rows[j] : j;
    if (v.acc[r] == 0.0) return;
    const bool ident = v.ident[r] != 0.0, pend = v.pend[r] != 0.0;
    if (ident && !pend) return;
    const int P = v.P, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + wv) * PGL_HM_ROWS;
    if (i0 >= P) return;
    const double hs = v.hscale[r];
    const double* __restrict__ g = v.g + (size_t)r * P;
    const double* __restrict__ U = v.U + (size_t)r * P * 3;
    const double* __restrict__ V = v.V + (size_t)r * P * 3;
    double* Hr = H + (size_t)r * P * ld;
    double acc[PGL_HM_ROWS], u[PGL_HM_ROWS][3];
#pragma unroll
    for (int k = 0; k < PGL_HM_ROWS; ++k) {
        acc[k] = 0.0;
        const int i = min(i0 + k, P - 1);
#pragma unroll
        for (int e = 0; e < 3; ++e) u[k][e] = pend ? U[(size_t)i * 3 + e] : 0.0;
    }
    for (int c = 2 * lane; c < ld; c += 128) {
        const bool in0 = c < P, in1 = c + 1 < P;
        const double g0 = in0 ? g[c] : 0.0, g1 = in1 ? g[c + 1] : 0.0;
        double va[3], vb[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            va[e] = (pend && in0) ? V[(size_t)c * 3 + e] : 0.0;
            vb[e] = (pend && in1) ? V[(size_t)(c + 1) * 3 + e] : 0.0;
        }
        double2 h[PGL_HM_ROWS];
#pragma unroll
        for (int k = 0; k < PGL_HM_ROWS; ++k) {
            const int i = i0 + k;
            if (ident || i >= P) h[k] = make_double2(i == c ? hs : 0.0, i == c + 1 ? hs : 0.0);
            else h[k] = *reinterpret_cast<const double2*>(Hr + (size_t)i * ld + c);
        }
#pragma unroll
        for (int k = 0; k < PGL_HM_ROWS; ++k) {
            const int i = i0 + k;
            if (pend) {
                h[k].x += u[k][0] * va[0] + u[k][1] * va[1] + u[k][2] * va[2];
                h[k].y += u[k][0] * vb[0] + u[k][1] * vb[1] + u[k][2] * vb[2];
                if (i < P) *reinterpret_cast<double2*>(Hr + (size_t)i * ld + c) = h[k];
            }
            acc[k] = fma(h[k].x, g0, fma(h[k].y, g1, acc[k]));
        }
    }
#pragma unroll
    for (int k = 0; k < PGL_HM_ROWS; ++k) {
        double a = acc[k];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0 && i0 + k < P) v.t[(size_t)r * P + i0 + k] = a;
    }
}

// The same product with the inverse Hessian kept IMPLICIT: H = hscale I + sum_{j < hk} U_j V_j^T over every update so far.
// The rank-3 factors share two vectors -- U_j = (c0 s, -rho Hy, -rho s), V_j = (s, s, Hy) -- so the history holds (s_j, Hy_j)
// [row][j][2][P] and (c0_j, rho_j) [row][j][2] (k_bfgs_update appends them), and
//   U_j V_j^T g = (c0 a - rho b) s_j + (-rho a) Hy_j,   a = s_j . g,  b = Hy_j . g.
// 4 hk P numbers per row and product instead of the 2 P^2 of the dense form -- less traffic while hk <= P / 2, which is
// where fits live (C3 converges in 22 iterations; at the C5 stress shape, P = 1220, all 225 iterations read 5 x less on
// average), and no P^2 memory (wide populations).  The driver's default.  Two kernels:
//   k_bfgs_hdots: ab[row][j] = the two coefficients, one wave per (row, j), fixed-order wave reduction;
//   k_bfgs_hcomb: t = hscale g + sum_j ab[j][0] s_j + ab[j][1] Hy_j for 64 components per block; wave w of 8 takes
//                 j = w, w + 8, ..., the eight partial sums are added in wave order.
__global__ __launch_bounds__(256) void k_bfgs_hdots(const BfgsView v, const int* __restrict__ rows,
                                                    const double* __restrict__ Wh, const double* __restrict__ cs, const int Kmax,
                                                    double* __restrict__ ab)
{
    const int jr = blockIdx.y, r = rows ? rows[jr] : jr;
    if (v.acc[r] == 0.0) return;
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6), P = v.P;
    if (j >= (int)v.hk[r]) return;
    const double* __restrict__ g = v.g + (size_t)r * P;
    const double* __restrict__ W = Wh + ((size_t)r * Kmax + j) * 2 * P;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    int i = lane;
    for (; i + 64 < P; i += 128) {
        const double g0 = g[i], g1 = g[i + 64];
        const double s0 = W[i], s1 = W[i + 64], h0 = W[P + i], h1 = W[P + i + 64];
        a0 = fma(s0, g0, a0);
        a1 = fma(s1, g1, a1);
        b0 = fma(h0, g0, b0);
        b1 = fma(h1, g1, b1);
    }
    if (i < P) {
        a0 = fma(W[i], g[i], a0);
        b0 = fma(W[P + i], g[i], b0);
    }
    double a = a0 + a1, b = b0 + b1;
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (lane == 0) {
        const size_t q = ((size_t)r * Kmax + j) * 2;
        const double c0 = cs[q], rho = cs[q + 1];
        ab[q] = c0 * a - rho * b;
        ab[q + 1] = -rho * a;
    }
}
__global__ __launch_bounds__(512) void k_bfgs_hcomb(const BfgsView v, const int* __restrict__ rows,
                                                    const double* __restrict__ Wh, const int Kmax, const double* __restrict__ ab)
{
    __shared__ double part[8][64];
    const int jr = blockIdx.y, r = rows ? rows[jr] : jr;
    if (v.acc[r] == 0.0) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, P = v.P, K = (int)v.hk[r];
    if (K == 0) return;                                      // (no history: k_bfgs_update uses t = hscale g)
    const int i = blockIdx.x * 64 + lane;
    const bool in = i < P;
    const double* __restrict__ W = Wh + (size_t)r * Kmax * 2 * P + (in ? i : 0);
    const double* __restrict__ c = ab + (size_t)r * Kmax * 2;
    double acc0 = 0.0, acc1 = 0.0;
    int j = w;
    for (; j + 8 < K; j += 16) {
        const double* u0 = W + (size_t)j * 2 * P;
        const double* u1 = W + (size_t)(j + 8) * 2 * P;
        const double s0 = u0[0], h0 = u0[P], s1 = u1[0], h1 = u1[P];
        acc0 = fma(s0, c[2 * j], acc0);
        acc0 = fma(h0, c[2 * j + 1], acc0);
        acc1 = fma(s1, c[2 * j + 16], acc1);
        acc1 = fma(h1, c[2 * j + 17], acc1);
    }
    if (j < K) {
        const double* u0 = W + (size_t)j * 2 * P;
        acc0 = fma(u0[0], c[2 * j], acc0);
        acc0 = fma(u0[P], c[2 * j + 1], acc0);
    }
    part[w][lane] = acc0 + acc1;
    __syncthreads();
    if (w == 0 && in) {
        double t = v.hscale[r] * v.g[(size_t)r * P + i];
        #pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][lane];
        v.t[(size_t)r * P + i] = t;
    }
}

// After the line-search step (and t = H g_new for the rows that moved): the rank-3 factors of the inverse-Hessian
// update  H_new = (I - rho s y^T) H (I - rho y s^T) + rho s s^T = H + U V^T,  H_new g_new, the next direction and the
// start of its line search, restart / freeze of stalled rows, convergence flags.  init_scaling != 0: the first update
// after a (re)start is preceded by H <- (s.y / y.y) I (Nocedal & Wright (6.20); not scipy's behaviour).
__device__ __forceinline__ void pgl_bfgs_update_row(const BfgsView& v, const int r, const double gtol, const int maxiter,
                                                    const int init_scaling, double* __restrict__ Wh,
                                                    double* __restrict__ cs, const int Kmax, double* red, const int tid)
{
    const int P = v.P;
    const size_t o = (size_t)r * P;
    const bool act = v.active[r] != 0.0, a = v.acc[r] != 0.0, u = v.upd[r] != 0.0, st = v.stall[r] != 0.0;
    if (!act || (!a && !st)) return;                         // finished, or in the middle of a line search
    double* U = v.U + o * 3;
    double* V = v.V + o * 3;
    bool ident = v.ident[r] != 0.0, pend = v.pend[r] != 0.0;
    double hs = v.hscale[r];
    double restarts = v.restarts[r], iters = v.iters[r];
    bool frozen = false, again = false, hist_add = false;
    if (a) {
        const bool lazy = ident && !pend;                    // H = hs * I: k_bfgs_hmul / k_bfgs_hcomb left t alone
        iters += 1.0;
        restarts = 0.0;
        if (u) {
            const double rho = v.rho[r];
            double sc = 1.0;
            if (init_scaling && lazy) {                      // H <- (s.y / y.y) I before the first update
                double yy = 0.0;
                for (int c = pgl_row_c0(tid, P); c < P; c += 256) yy = fma(v.y[o + c], v.y[o + c], yy);
                yy = pgl_blk_sum(yy, red);
                const double gam = (1.0 / rho) / yy;
                if (gam > 0.0 && gam - gam == 0.0) { sc = gam / hs; hs = gam; }
            }
            double yHy = 0.0, vg0 = 0.0, vg2 = 0.0;
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
                const double tc = lazy ? hs * v.g[o + c] : v.t[o + c];
                const double Hy = tc - sc * v.Hg[o + c];
                yHy = fma(v.y[o + c], Hy, yHy);
                vg0 = fma(v.s[o + c], v.g[o + c], vg0);
                vg2 = fma(Hy, v.g[o + c], vg2);
            }
            yHy = pgl_blk_sum(yHy, red);
            vg0 = pgl_blk_sum(vg0, red);
            vg2 = pgl_blk_sum(vg2, red);
            const double c0 = (1.0 + rho * yHy) * rho;
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
                const double tc = lazy ? hs * v.g[o + c] : v.t[o + c];
                const double s = v.s[o + c], Hy = tc - sc * v.Hg[o + c];
                const double u0 = c0 * s, u1 = -rho * Hy, u2 = -rho * s;
                v.Hg[o + c] = tc + (u0 * vg0 + u1 * vg0 + u2 * vg2);          // H_new g_new
                if (Wh) {                                    // implicit form: the update joins the history
                    const size_t hq = ((size_t)r * Kmax + (size_t)v.hk[r]) * 2 * P + c;
                    Wh[hq] = s;
                    Wh[hq + P] = Hy;
                } else {
                    U[3 * c] = u0; U[3 * c + 1] = u1; U[3 * c + 2] = u2;
                    V[3 * c] = s; V[3 * c + 1] = s; V[3 * c + 2] = Hy;
                }
            }
            if (Wh && tid == 0) {
                const size_t cq = ((size_t)r * Kmax + (size_t)v.hk[r]) * 2;
                cs[cq] = c0;
                cs[cq + 1] = rho;
            }
            ident = Wh ? false : lazy;                       // dense: still not materialised -- hs * I + U V^T at the next pass
            pend = Wh ? false : true;
            hist_add = Wh != nullptr;
        } else {
            for (int c = pgl_row_c0(tid, P); c < P; c += 256) v.Hg[o + c] = lazy ? hs * v.g[o + c] : v.t[o + c];
            ident = lazy;
            pend = false;
        }
        __syncthreads();
    } else {                                                 // stalled line search: restart once, then freeze
        again = restarts == 0.0;
        if (again) restarts = 1.0;
        else frozen = true;
    }
    double sl = 0.0, gg = 0.0, gmax = 0.0;
    for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
        const double gc = v.g[o + c];
        sl = fma(-v.Hg[o + c], gc, sl);
        gg = fma(gc, gc, gg);
        gmax = fmax(gmax, fabs(gc));
    }
    sl = pgl_blk_sum(sl, red);
    gg = pgl_blk_sum(gg, red);
    gmax = pgl_blk_max(gmax, red);
    const bool newls = a || again;
    const bool reset = newls && (again || !(sl < 0.0));      // restart / not a descent direction: H = I
    if (newls)
        for (int c = pgl_row_c0(tid, P); c < P; c += 256) {
            const double gc = v.g[o + c];
            if (reset) v.Hg[o + c] = gc;
            v.p[o + c] = reset ? -gc : -v.Hg[o + c];
        }
    if (tid == 0) {
        v.iters[r] = iters;
        v.restarts[r] = restarts;
        if (frozen) v.frozen[r] = 1.0;
        if (reset) { ident = true; pend = false; hs = 1.0; }
        v.hk[r] = reset ? 0.0 : v.hk[r] + (hist_add ? 1.0 : 0.0);
        v.ident[r] = ident ? 1.0 : 0.0;
        v.pend[r] = pend ? 1.0 : 0.0;
        v.hscale[r] = hs;
        const bool go = !frozen && gmax > gtol && iters < (double)maxiter;
        if (newls && go) {
            const double slope = reset ? -gg : sl;
            const double f = v.f[r];
            const double fprev = again ? f + sqrt(gg) / 2.0 : v.fprev[r];
            v.fprev[r] = fprev;
            v.slope[r] = slope;
            PglLs s;
            const double a0 = pgl_ls_first_step(f, fprev, slope);
            pgl_ls_start(&s, a0, f, slope, PGL_LS_FTOL, PGL_LS_STPMIN, PGL_LS_STPMAX);
            pgl_ls_store(v, r, &s);
            v.alpha[r] = a0;
            v.fb[r] = f;
        }
        v.active[r] = go ? 1.0 : 0.0;
        v.acc[r] = 0.0;                                      // cleared for the next launch
        v.upd[r] = 0.0;
        v.stall[r] = 0.0;
    }
}

// t = H g of ONE row from its update history inside a block of NT threads (NT / 64 waves): the two kernels above in one
// workgroup -- wave w takes the dots of the updates j = w, w + NT / 64, ... exactly as a wave of k_bfgs_hdots does, then
// NT / 512 groups of eight waves each combine 64 components exactly as a block of k_bfgs_hcomb does: the same numbers.
// The merged iteration kernel uses it while the history of a row is short (4 hk P numbers through one workgroup).
template <int NT>
__device__ __forceinline__ void pgl_bfgs_hist_row(const BfgsView& v, const int r, const double* __restrict__ Wh,
                                                  const double* __restrict__ cs, const int Kmax, double* __restrict__ ab,
                                                  double (*part)[64], const int tid)
{
    if (v.acc[r] == 0.0) return;
    const int P = v.P, K = (int)v.hk[r];
    if (K == 0) return;                                      // (no history: the update uses t = hscale g)
    const int lane = tid & 63, wv = tid >> 6;
    const double* __restrict__ g = v.g + (size_t)r * P;
    for (int j = wv; j < K; j += NT / 64) {
        const double* __restrict__ W = Wh + ((size_t)r * Kmax + j) * 2 * P;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        int i = lane;
        for (; i + 64 < P; i += 128) {
            const double g0 = g[i], g1 = g[i + 64];
            const double s0 = W[i], s1 = W[i + 64], h0 = W[P + i], h1 = W[P + i + 64];
            a0 = fma(s0, g0, a0);
            a1 = fma(s1, g1, a1);
            b0 = fma(h0, g0, b0);
            b1 = fma(h1, g1, b1);
        }
        if (i < P) {
            a0 = fma(W[i], g[i], a0);
            b0 = fma(W[P + i], g[i], b0);
        }
        double a = a0 + a1, b = b0 + b1;
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        if (lane == 0) {
            const size_t q = ((size_t)r * Kmax + j) * 2;
            const double c0 = cs[q], rho = cs[q + 1];
            ab[q] = c0 * a - rho * b;
            ab[q + 1] = -rho * a;
        }
    }
    __syncthreads();                                         // the coefficients are in memory for the whole workgroup
    constexpr int NVB = (NT >= 512) ? NT / 512 : 1;          // groups of eight waves
    constexpr int WPG = (NT >= 512) ? 8 : NT / 64;           // (a 256-thread block: one group of four -- not used)
    static_assert(NT >= 512 && NT % 512 == 0, "pgl_bfgs_hist_row: whole groups of eight waves");
    const int vb = wv / WPG, w = wv % WPG;
    const int nblk = (P + 63) / 64;
    const double* __restrict__ c = ab + (size_t)r * Kmax * 2;
    for (int b0 = 0; b0 < nblk; b0 += NVB) {
        const int blk = b0 + vb;
        const int i = blk * 64 + lane;
        const bool in = blk < nblk && i < P;
        const double* __restrict__ W = Wh + (size_t)r * Kmax * 2 * P + (in ? i : 0);
        double acc0 = 0.0, acc1 = 0.0;
        int j = w;
        for (; j + 8 < K; j += 16) {
            const double* u0 = W + (size_t)j * 2 * P;
            const double* u1 = W + (size_t)(j + 8) * 2 * P;
            const double s0 = u0[0], h0 = u0[P], s1 = u1[0], h1 = u1[P];
            acc0 = fma(s0, c[2 * j], acc0);
            acc0 = fma(h0, c[2 * j + 1], acc0);
            acc1 = fma(s1, c[2 * j + 16], acc1);
            acc1 = fma(h1, c[2 * j + 17], acc1);
        }
        if (j < K) {
            const double* u0 = W + (size_t)j * 2 * P;
            acc0 = fma(u0[0], c[2 * j], acc0);
            acc0 = fma(u0[P], c[2 * j + 1], acc0);
        }
        part[wv][lane] = acc0 + acc1;
        __syncthreads();
        if (w == 0 && in) {
            double t = v.hscale[r] * v.g[(size_t)r * P + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) t += part[vb * 8 + q][lane];
            v.t[(size_t)r * P + i] = t;
        }
        __syncthreads();
    }
}

// One iteration of the lock-step optimiser for the listed rows, one workgroup per row: the phases an evaluation is followed
// by, selected by `phases` so that the same code serves the one-launch form and the split form around the multi-workgroup
// inverse-Hessian kernels (k_bfgs_hdots + k_bfgs_hcomb for long histories, k_bfgs_hmul for dense matrices):
//   PGL_STEP_LS     priors + NaN rules on the evaluation's (ll, grad) (have_prior; else f, g arrive final), line-search step
//   PGL_STEP_HIST   t = H g from the update history of a row that has just taken a step (NT >= 512)
//   PGL_STEP_UPDATE BFGS update / restart / freeze / convergence, next direction and first step; then the row's next trial
//                   point into Xt_next[pos_next[r]] (the list of the NEXT launch: pos_next null = same positions) and its
//                   active flag into flags_out[r] (host-visible memory: the driver reads it without a copy kernel)
// The rows are the evaluation's list (rows[j], null: j); Xt / ft / gt are indexed by list position.
#define PGL_STEP_LS 1
#define PGL_STEP_HIST 2
#define PGL_STEP_UPDATE 4
struct BfgsStepArgs {
    const int* rows;
    const double* Xt;
    double* ft;
    double* gt;
    BfgsPrior q;
    int have_prior, max_trials, maxiter, init_scaling, Kmax, phases;
    double gtol;
    double* Wh;
    double* cs;
    double* ab;
    const int* pos_next;
    double* Xt_next;
    double* flags_out;
};
template <int NT>
__global__ __launch_bounds__(NT) void k_bfgs_step(const BfgsView v, const BfgsStepArgs a)
{
    __shared__ double red[4];
    __shared__ int dec[3];
    __shared__ double part[(NT >= 512) ? NT / 64 : 1][64];
    const int j = blockIdx.x, r = a.rows ? a.rows[j] : j, tid = threadIdx.x, P = v.P;
    if (a.phases & PGL_STEP_LS) {
        if (a.have_prior && v.active[r] != 0.0)
            pgl_bfgs_objective_row(P, a.Xt + (size_t)j * P, a.gt + (size_t)j * P, a.ft + j, a.q, red, tid);
        __syncthreads();                                     // f, g of the trial are in memory for the whole workgroup
        pgl_bfgs_linesearch_row(v, r, a.Xt + (size_t)j * P, a.ft + j, a.gt + (size_t)j * P, a.max_trials, red, dec, tid);
        __syncthreads();
    }
    if constexpr (NT >= 512) {
        if ((a.phases & PGL_STEP_HIST) && a.Wh) {
            pgl_bfgs_hist_row<NT>(v, r, a.Wh, a.cs, a.Kmax, a.ab, part, tid);
            __syncthreads();
        }
    }
    if (a.phases & PGL_STEP_UPDATE) {
        pgl_bfgs_update_row(v, r, a.gtol, a.maxiter, a.init_scaling, a.Wh, a.cs, a.Kmax, red, tid);
        __syncthreads();
        if (a.Xt_next) {
            const int jn = a.pos_next ? a.pos_next[r] : j;
            if (jn >= 0) {
                const double al = v.alpha[r];
                for (int c = pgl_row_c0(tid, P); c < P; c += 256)
                    a.Xt_next[(size_t)jn * P + c] = fma(al, v.p[(size_t)r * P + c], v.X[(size_t)r * P + c]);
            }
        }
        if (a.flags_out && tid == 0)
            __hip_atomic_store(a.flags_out + r, v.active[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
