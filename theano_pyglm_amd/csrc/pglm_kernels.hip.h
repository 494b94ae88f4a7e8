// Device kernels of the population-GLM likelihood library (gfx950 / CDNA4 only).
//
// Reference arithmetic (slinderman/theano_pyglm):
//   features   fS[t,n',b] = sum_{tau=1..R} S[t-tau,n'] ibasis[tau-1,b]   pyglm/utils/basis.py:201-236
//   currents   x[t,n] = bias_n + fstim[t,:].wstim_n + sum_{n',b} fS[t,n',b] w_n[n',b] Weff[n',n]
//                                                                         pyglm/glm.py:31-45, impulse.py:58
//   likelihood ll_n = sum_t ( -dt*lam + log(lam)*S[t,n] ),  lam = nlin(x)   pyglm/glm.py:52, nlin.py:25,43
//   gradient   T.grad(glm.ll, [bias, w_stim, w_ir])                        pyglm/inference/coord_descent.py:27-30
//
// Design: the feature matrix F (nT x K, K = N*B + Dstim) is never
// materialised in HBM.  A workgroup walks a chunk of 16-row time tiles; for each
// tile it rebuilds the F tile in LDS from the sparse spike-event list (CSR per
// presynaptic neuron) and the LDS-resident basis table, then every wave (one 16-wide
// post-synaptic tile each) runs  X = F.Wmat  with v_mfma_f64_16x16x4_f64, turns X
// into (ll, r = dll/dx) in registers, and immediately accumulates  G += F^T.r  with a
// second MFMA pass whose B operand IS the forward accumulator layout (no shuffle).
// G (K x 16 per wave) stays in registers for the whole chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4_t __attribute__((ext_vector_type(4)));

#define PGL_CAP 16          // staged spike events per presynaptic neuron and tile
#define PGL_MAXB 8

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
    int RP;                              // padded basis-table length (>= R+32, RP % 32 == 6)
    double* __restrict__ Gpart;          // [nChunks][nPT][KT][4][64]
    double* __restrict__ llpart;         // [nChunks][nPT][64]
    double* __restrict__ gbpart;         // [nChunks][nPT][64]
    int want_grad;
    int dbg;                             // timing ablation bits (results invalid when != 0)
};

__device__ __forceinline__ double pgl_softplus_parts(double x, double& sig, double& loglam)
{
    // stable log(1+exp(x)) (nlin.py:43); sigmoid and log(lam) share the exp
    const double e = exp(-fabs(x));
    const double lam = fmax(x, 0.0) + log1p(e);
    const double inv = 1.0 / (1.0 + e);
    sig = (x >= 0.0) ? inv : e * inv;
    loglam = log(lam);
    return lam;
}

// ---------------------------------------------------------------------------
// Feature generation for one 16-row time tile: F[t][n'*B+b], t in [t0,t0+16).
// One thread owns a feature column (n',b) and keeps its 16 rows in registers; every
// event (s,c) of n' in the tile's window adds c*phi_b[t0+t-s-1] to row t.  The basis
// table is zero-padded by 16 taps on both sides, so no per-row validity test is
// needed (the window table only admits events with s in [t0-R, t0+14]); an even- and
// an odd-shifted copy keep every 16-tap slice 16-byte aligned for ds_read_b128.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gen_accum16(double (&acc)[16], const int2 e, const int t0,
                                            const double* __restrict__ phiE,
                                            const double* __restrict__ phiO, const int boff)
{
    const int base = t0 - e.x - 1 + 16;               // padded index of row 0's lag, >= 1
    const double c = (double)e.y;
    const double* tab = (base & 1) ? (phiO + boff + base - 1) : (phiE + boff + base);
    const double2* t2 = reinterpret_cast<const double2*>(tab);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double2 v = t2[q];
        acc[2 * q] = fma(c, v.x, acc[2 * q]);
        acc[2 * q + 1] = fma(c, v.y, acc[2 * q + 1]);
    }
}

template <typename FT>
__device__ __forceinline__ void gen_cols(FT* __restrict__ Fs, const int rsf,
                                         const double* __restrict__ phiE,
                                         const double* __restrict__ phiO, const int RP,
                                         const int2* __restrict__ s_spk,
                                         const int* __restrict__ s_lo,
                                         const int* __restrict__ s_cnt,
                                         const int2* __restrict__ spk, const int t0, const int B,
                                         const int Kimp, const int tid, const int nthr)
{
    for (int col = tid; col < Kimp; col += nthr) {
        const int np = col / B;
        const int b = col - np * B;
        const int cnt = s_cnt[np];
        const int boff = b * RP;
        double acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0;
        // two events per trip (the second is a zero-weight dummy when cnt is odd) so that
        // 16 independent ds_read_b128 are in flight before the FMAs
        const int2 dummy = make_int2(t0, 0);
        if (cnt <= PGL_CAP) {
            const int2* sp = s_spk + np * PGL_CAP;
            for (int j = 0; j < cnt; j += 2) {
                const int2 e0 = sp[j];
                const int2 e1 = (j + 1 < cnt) ? sp[j + 1] : dummy;
                gen_accum16(acc, e0, t0, phiE, phiO, boff);
                gen_accum16(acc, e1, t0, phiE, phiO, boff);
            }
        } else {
            const int2* sp = spk + s_lo[np];
            for (int j = 0; j < cnt; ++j) gen_accum16(acc, sp[j], t0, phiE, phiO, boff);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) Fs[t * rsf + col] = (FT)acc[t];
    }
}

// ---------------------------------------------------------------------------
// The fused ll + grad kernel.  One wave = one 16-wide post-synaptic tile.
// ---------------------------------------------------------------------------
template <int KT, typename FT>
__global__ __launch_bounds__(256, 1) void k_fused_ll_grad(const FusedParams p)
{
    constexpr int TT = 16;
    constexpr int KS = KT * 4;            // k-steps of 4 in the forward pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR)
    const int wpb = nthr >> 6;
    const int nPB = (p.nPT + wpb - 1) / wpb;
    const int pb = blockIdx.x % nPB;
    const int chunk = blockIdx.x / nPB;
    const int pt = pb * wpb + wave;
    const bool active = pt < p.nPT;

    const int N = p.N, B = p.B, R = p.R, rsf = p.rsf;
    // LDS carve (all offsets multiples of 16)
    FT* Fs = reinterpret_cast<FT*>(smem);
    size_t off = ((size_t)TT * rsf * sizeof(FT) + 15) & ~(size_t)15;
    const int RP = p.RP;
    double* phiE = reinterpret_cast<double*>(smem + off);
    double* phiO = phiE + (size_t)B * RP;
    off += (((size_t)2 * B * RP * 8) + 15) & ~(size_t)15;
    int2* s_spk = reinterpret_cast<int2*>(smem + off);
    off += (size_t)N * PGL_CAP * 8;
    int* s_lo = reinterpret_cast<int*>(smem + off);
    off += (((size_t)N * 4) + 15) & ~(size_t)15;
    int* s_cnt = reinterpret_cast<int*>(smem + off);

    // one-time: basis table, zero the F tile (padding columns stay zero forever)
    for (int i = tid; i < B * RP; i += nthr) {
        const int b = i / RP, k = i - b * RP;
        phiE[i] = (k >= 16 && k < 16 + R) ? p.phi[b * R + k - 16] : 0.0;
        phiO[i] = (k + 1 >= 16 && k + 1 < 16 + R) ? p.phi[b * R + k + 1 - 16] : 0.0;
    }
    for (int i = tid; i < TT * rsf; i += nthr) Fs[i] = (FT)0;

    d4_t G[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) G[kt] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double ll_acc = 0.0, gb_acc = 0.0;

    const int col = lane & 15;
    const int grp = lane >> 4;
    const int nloc = pt * 16 + col;                     // neuron index inside [n_lo,n_hi)
    const bool valid_n = active && (nloc < p.npost);
    const int nglob = p.n_lo + (valid_n ? nloc : 0);
    const double bias_l = valid_n ? p.bias[nloc] : 0.0;
    // wave-uniform base (SGPR pair) + lane offset: global_load saddr form, no per-step VGPR addresses
    const double* __restrict__ wrow = p.Wfrag + ((size_t)(active ? pt : 0) * KS) * 64;

    const int tile_beg = chunk * p.tilesPerChunk;
    int tile_end = tile_beg + p.tilesPerChunk;
    if (tile_end > p.nTiles) tile_end = p.nTiles;

    __syncthreads();

    for (int tile = tile_beg; tile < tile_end; ++tile) {
        const int t0 = tile * TT;
        // ---- phase A: event window of every presynaptic neuron, staged to LDS ----
        if (tid < N && !(p.dbg & 32)) {
            const int lo = p.wlo[(size_t)tile * N + tid];
            const int hi = p.whi[(size_t)tile * N + tid];
            s_lo[tid] = lo;
            s_cnt[tid] = hi - lo;
        }
        __syncthreads();
        if (!(p.dbg & 2))
        for (int id = tid; id < N * PGL_CAP; id += nthr) {
            const int np = id / PGL_CAP;
            const int sl = id % PGL_CAP;
            const int cnt = s_cnt[np];
            if (cnt <= PGL_CAP && sl < cnt) s_spk[id] = p.spk[s_lo[np] + sl];
        }
        // dense stimulus feature columns
        if (p.Dstim > 0) {
            for (int id = tid; id < TT * p.Dstim; id += nthr) {
                const int t = id / p.Dstim;
                const int j = id % p.Dstim;
                const long long tg = (long long)t0 + t;
                Fs[t * rsf + p.Kimp + j] = (FT)((tg < p.nT) ? p.fstim[tg * p.Dstim + j] : 0.0);
            }
        }
        __syncthreads();
        // ---- phase B: F tile from events ----
        if (!(p.dbg & 1))
            gen_cols<FT>(Fs, rsf, phiE, phiO, RP, s_spk, s_lo, s_cnt, p.spk, t0, B, p.Kimp, tid, nthr);
        __syncthreads();

        if (active) {
            // post-synaptic counts for the epilogue (issued early, used after the MFMAs)
            double sc[4];
            bool vt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long tg = (long long)t0 + grp + 4 * r;
                vt[r] = valid_n && (tg < p.nT);
                const long long tc = (tg < p.nT) ? tg : (p.nT - 1);      // clamped: branch-free load
                sc[r] = (double)p.S[tc * N + nglob];
            }
            // ---- forward: X(16x16) = F(16xK) . Wmat(Kx16) ----
            // Flat list of KS MFMAs.  The Wmat fragments (L2 -> VGPR) are fetched PW steps
            // ahead and the F fragments (LDS -> VGPR) PA steps ahead through register rings
            // whose indices are static after unrolling; two accumulators break the
            // dependent-accumulate chain.
            d4_t acc0 = (d4_t){0.0, 0.0, 0.0, 0.0};
            d4_t acc1 = (d4_t){0.0, 0.0, 0.0, 0.0};
            {
                const FT* fa = Fs + col * rsf + grp;  // A[i=t=lane&15][k=lane>>4]
                // opaque SGPR copy of the wave-uniform fragment base: keeps the compiler from
                // hoisting 160 per-step VGPR addresses out of the tile loop (they spill) and
                // selects the saddr form  global_load_dwordx2 v, v_lane8, s[base] offset:imm
                const double* wr_s = wrow;
                asm volatile("" : "+s"(wr_s));
                constexpr int PW = (KS < 12) ? KS : 12;
                constexpr int PA = 4;
                double wr[PW], ar[PA];
#pragma unroll
                for (int s = 0; s < PW; ++s) wr[s] = wr_s[s * 64 + lane];
#pragma unroll
                for (int s = 0; s < PA; ++s) ar[s] = (double)fa[4 * s];
                if (!(p.dbg & 8))
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ar[s % PA];
                    const double b = wr[s % PW];
                    if (s + PA < KS) ar[s % PA] = (double)fa[4 * (s + PA)];
                    if (s + PW < KS) wr[s % PW] = wr_s[(s + PW) * 64 + lane];
                    if (s & 1)
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                    else
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- epilogue: x -> (ll, r).  D layout: row t = grp + 4r, col = lane&15 ----
            double rr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double x = acc0[r] + acc1[r] + bias_l;
                double term, res;
                if (p.dbg & 4) {
                    term = x * sc[r];
                    res = x - sc[r];
                } else if (p.nlin == 1) {
                    double sig, loglam;
                    const double lam = pgl_softplus_parts(x, sig, loglam);
                    term = -p.dt * lam + loglam * sc[r];
                    res = (-p.dt + sc[r] / lam) * sig;
                } else {
                    const double lam = exp(x);
                    term = -p.dt * lam + x * sc[r];
                    res = -p.dt * lam + sc[r];
                }
                ll_acc += vt[r] ? term : 0.0;
                rr[r] = vt[r] ? res : 0.0;
                gb_acc += rr[r];
            }
            // ---- backward: G(Kx16) += F^T(Kx16) . r(16x16);  B operand of k-step j is rr[j] ----
            // Flat list of 4*KT MFMAs (step s: time k-step j = s / KT, feature tile kt = s % KT),
            // F^T fragments fetched from LDS PD steps ahead.
            if (p.want_grad && !(p.dbg & 16)) {
                const FT* fb = Fs + grp * rsf + col;  // A[i=k=lane&15][kk=lane>>4] = F[4j+kk][16kt+i]
                constexpr int PD = 4;
                constexpr int NS = 4 * KT;
                double ar[PD];
#pragma unroll
                for (int s = 0; s < PD; ++s) ar[s] = (double)fb[(4 * (s / KT)) * rsf + 16 * (s % KT)];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double a = ar[s % PD];
                    if (s + PD < NS)
                        ar[s % PD] = (double)fb[(4 * ((s + PD) / KT)) * rsf + 16 * ((s + PD) % KT)];
                    G[s % KT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rr[s / KT], G[s % KT], 0, 0, 0);
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
    }

    if (active) {
        const size_t slot = (size_t)chunk * p.nPT + pt;
        p.llpart[slot * 64 + lane] = ll_acc;
        p.gbpart[slot * 64 + lane] = gb_acc;
        if (p.want_grad) {
            double* gp = p.Gpart + slot * (size_t)KT * 256 + lane;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gp[(kt * 4 + r) * 64] = G[kt][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// prep: Wmat in MFMA B-fragment order + bias vector
//   Wfrag[pt][ks][lane] = Wmat[k = 4ks + (lane>>4)][n = 16pt + (lane&15)]
// ---------------------------------------------------------------------------
__global__ void k_prep_w(const double* __restrict__ theta, const double* __restrict__ Weff,
                         double* __restrict__ Wfrag, double* __restrict__ bias, int N, int B,
                         int Dstim, int Kimp, int Ktot, int KS, int n_lo, int npost, int nPT)
{
    const int P = 1 + Dstim + Kimp;
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
                const int npre = k / B;
                v = theta[(size_t)n * P + 1 + Dstim + k] * Weff[(size_t)npre * N + (n_lo + n)];
            } else {
                v = theta[(size_t)n * P + 1 + (k - Kimp)];
            }
        }
        Wfrag[i] = v;
    }
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nPT * 16; n += gridDim.x * blockDim.x)
        bias[n] = (n < npost) ? theta[(size_t)n * P] : 0.0;
}

// ---------------------------------------------------------------------------
// finalize: deterministic reduction of the per-chunk partials, Weff chain rule,
// scatter into the (npost, P) gradient layout
// ---------------------------------------------------------------------------
__global__ void k_finalize(const double* __restrict__ Gpart, const double* __restrict__ llpart,
                           const double* __restrict__ gbpart, const double* __restrict__ Weff,
                           double* __restrict__ ll_out, double* __restrict__ grad_out, int N, int B,
                           int Dstim, int Kimp, int Ktot, int KT, int n_lo, int npost, int nPT,
                           int nChunks)
{
    const int P = 1 + Dstim + Kimp;
    const long long nfrag = (long long)nPT * KT * 256;
    const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (grad_out != nullptr && gid < nfrag) {
        const int lane = (int)(gid & 63);
        const int r = (int)((gid >> 6) & 3);
        const int kt = (int)((gid >> 8) % KT);
        const int pt = (int)((gid >> 8) / KT);
        const int k = 16 * kt + (lane >> 4) + 4 * r;
        const int n = 16 * pt + (lane & 15);
        if (n < npost && k < Ktot) {
            double s = 0.0;
            for (int c = 0; c < nChunks; ++c)
                s += Gpart[((size_t)c * nPT + pt) * (size_t)KT * 256 + (size_t)(kt * 4 + r) * 64 + lane];
            if (k < Kimp) {
                const int npre = k / B;
                grad_out[(size_t)n * P + 1 + Dstim + k] = s * Weff[(size_t)npre * N + (n_lo + n)];
            } else {
                grad_out[(size_t)n * P + 1 + (k - Kimp)] = s;
            }
        }
    }
    if (gid < npost) {
        const int n = (int)gid;
        const int pt = n >> 4, col = n & 15;
        double sl = 0.0, sg = 0.0;
        for (int c = 0; c < nChunks; ++c) {
            const size_t base = ((size_t)c * nPT + pt) * 64;
            for (int g = 0; g < 4; ++g) {
                sl += llpart[base + g * 16 + col];
                sg += gbpart[base + g * 16 + col];
            }
        }
        ll_out[n] = sl;
        if (grad_out != nullptr) grad_out[(size_t)n * P] = sg;
    }
}

// ---------------------------------------------------------------------------
// direct-form helpers (not MFMA): features, impulse currents, state, MCMC inner ll
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
//   part[block][k] = sum_t (-dt*lam + log(lam)*S_T[t])
#define PGL_KMAX 16
__global__ __launch_bounds__(256) void k_ll_current(const double* __restrict__ base,
                                                    const double* __restrict__ stim,
                                                    const double* __restrict__ colv,
                                                    const uint8_t* __restrict__ Sn, double bias,
                                                    double aw_cur, const double* __restrict__ w,
                                                    int K, int nlin, double dt, long long nT,
                                                    double* __restrict__ part)
{
    __shared__ double red[4][PGL_KMAX];
    double wk[PGL_KMAX], acc[PGL_KMAX];
#pragma unroll
    for (int k = 0; k < PGL_KMAX; ++k) {
        wk[k] = (k < K) ? w[k] : 0.0;
        acc[k] = 0.0;
    }
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nT;
         t += (long long)gridDim.x * blockDim.x) {
        const double c = colv[t];
        const double x0 = bias + (stim ? stim[t] : 0.0) + base[t] - aw_cur * c;
        const double s = (double)Sn[t];
#pragma unroll
        for (int k = 0; k < PGL_KMAX; ++k) {
            if (k < K) {
                const double x = fma(wk[k], c, x0);
                double term;
                if (nlin == 1) {
                    double sig, loglam;
                    const double lam = pgl_softplus_parts(x, sig, loglam);
                    term = -dt * lam + loglam * s;
                } else {
                    term = -dt * exp(x) + x * s;
                }
                acc[k] += term;
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

__global__ void k_reduce_parts(const double* __restrict__ part, int nblocks, int K,
                               double* __restrict__ out)
{
    const int k = threadIdx.x;
    if (k < K) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * PGL_KMAX + k];
        out[k] = s;
    }
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
