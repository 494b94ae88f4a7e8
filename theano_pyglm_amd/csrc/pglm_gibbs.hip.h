// batched collapsed-Gibbs columns: k_gibbs_*
// Part of pglm_kernels.hip.h (included from there, in order; one translation unit).
#pragma once
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
