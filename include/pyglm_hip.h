/*
 * pyglm_hip.h -- C ABI of the MI355X (gfx950) population-GLM likelihood library.
 *
 * The reference (slinderman/theano_pyglm) has no FFI: its operator boundary is
 *   seval(expr, syms, vals)            pyglm/utils/theano_func_wrapper.py:12-51
 * evaluated on Theano shared variables filled by
 *   Glm.set_data / Population.set_data pyglm/glm.py:99-110, pyglm/population.py:223-231
 * Each entry point below names the reference expression(s) it replaces.  All
 * pointers are plain host pointers unless the name ends in `_dev`; the caller
 * owns host buffers, the library owns device buffers.  Every function returns
 * 0 on success and a negative code on failure; pgl_last_error() gives the text.
 * A handle is bound to one GPU; calls on one handle must be serialised by the
 * caller (like the reference's module-global _func_cache / shared variables).
 *
 * Flat feature-weight layout ("theta", one row per post-synaptic neuron):
 *     theta[0]                      bias                     (bias.py:32)
 *     theta[1 .. 1+Dstim)           stimulus feature weights (bkgd.py:81 / 227: w_stim,
 *                                   for SpatiotemporalStimulus vec(w_t (x) w_x), bkgd.py:214-220)
 *     theta[1+Dstim .. 1+Dstim+N*B) impulse weights w[n_pre*B + b]
 *                                   (impulse.py:58 w_ir; for DirichletImpulses beta, impulse.py:286-308)
 *   P = 1 + Dstim + N*B.  Gradients come back in the same layout (chain rules
 *   through w_t (x) w_x or |g|/sum|g| are applied by the host mirror).
 * Weff is the (N x N) row-major matrix A[n_pre,n_post]*W[n_pre,n_post] (glm.py:31-37).
 */
#ifndef PYGLM_HIP_H
#define PYGLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgl_context* pgl_handle;

#define PGL_OK 0
#define PGL_ERR_ARG (-1)       /* bad argument */
#define PGL_ERR_HIP (-2)       /* HIP runtime error (no device, OOM, launch failure) */
#define PGL_ERR_STATE (-3)     /* call order (e.g. ll before set_spikes) */
#define PGL_ERR_UNSUPPORTED (-4) /* shape outside what the kernels were built for */

#define PGL_NLIN_EXP 0         /* nlin.py:25 */
#define PGL_NLIN_EXPLINEAR 1   /* nlin.py:43 */

/* flags for pgl_set_option */
#define PGL_OPT_FEATURE_F32 1  /* 0 (default): f64 features.  1: the in-kernel-feature kernels stage the feature tile in LDS as f32.
                                * 2: reduced-traffic mode for narrow shards -- the RESIDENT feature blocks of the one-post-tile
                                * kernel (a shard of <= 16 neurons against a 400..640-column row: north star's neuron split at
                                * 8 GPUs) are STORED as f32 and widened to f64 on their way into LDS; every arithmetic operation
                                * stays f64, only the stored feature is rounded (2^-24 relative).  Other shapes ignore it. */
#define PGL_OPT_NCHUNKS 2      /* override the number of time chunks (0 = auto) */
#define PGL_OPT_KERNEL 3       /* 0 = auto: two-pass kernel on resident tiles for >= 65 post-synaptic neurons per call;
                                * below that the K-split kernel, on resident feature tiles (6) when the
                                * feature row is short enough for two LDS step buffers (N*B + Dstim up to
                                * ~320-450 columns depending on the post block), else with in-kernel
                                * feature generation (2); rows of <= 320 columns and <= 64 post neurons: one wave per
                                * post tile without any K split (7);  6 / 7 = force those kernels when they fit;
                                * 2 = force the K-split kernel; 3 = force the two-pass kernel with
                                * on-the-fly features; 4 = force the two-pass kernel on resident feature
                                * tiles.  Auto uses 4's kernel (k_fused5) when the call covers >= 65
                                * post-synaptic neurons and the device can hold the tiles
                                * (nT/16 * 2 * ~41 KB at K = 640: 3.1 GB for nT = 600 000) */

#define PGL_OPT_GIBBS_KERNEL 4 /* pgl_gibbs_ll_cols with the explinear nonlinearity: 0 = auto (regime-split kernels: single
                                * precision for the log1p(exp(-|x|)) term where |x| >= 12, compacted f64 elsewhere (log1p
                                * through a 32-interval table: < 2e-16 absolute),
                                * spike terms from the event lists); 1 = the all-f64 one-thread-per-(column, weight)
                                * kernel (always used for the exp nonlinearity) */
#define PGL_OPT_EPI_F64 5      /* 1 = all-f64 rate epilogue of the fused ll+grad kernels.  Default 0: in waves whose currents
                                * are all > 12 the exp(-x) < 6.2e-6 inside softplus / sigmoid comes from the single-precision
                                * hardware exp (rate and residual within 5e-13 relative of the all-f64 form) */

#define PGL_OPT_TIMING 6       /* HIP events around every n-th evaluation (pgl_last_timing / pgl_timing_summary): 1 = every call
                                * (default), n > 1 = a sample, 0 = none.  An event between two kernels of a stream costs ~6 us of
                                * GPU idle time; loops that queue evaluations back to back (optimisers, bench.py) sample or
                                * switch the events off */
#define PGL_OPT_BFGS_MERGE 7   /* pgl_bfgs_step_dev: a whole optimiser iteration is ONE row kernel while hk_bound * P <= value
                                * (doubles of update history per row that one workgroup walks; default 65536), else the split
                                * form with the multi-workgroup history kernels; 0 = always split */

/* Development switches (not part of the drop-in surface; results stay valid unless stated): 95 = 2 keeps the narrow post
 * blocks of a wide population off the one-image-buffer form of k_fused6 and the block-ring kernel k_fused8 (they run on
 * k_fused2); 97 = waves per block of
 * the partial reduction; 98 = post tiles per workgroup of the K-split kernels; 99 = kernel-internal ablation bits of the
 * fused / Gibbs kernels (!= 0 invalidates the results, except bit 0x1000: event-window pair currents although all columns
 * share the presynaptic neuron, and bits 8-11: forced sub-block count of k_gibbs_rate_cols). */

const char* pgl_last_error(void);
int pgl_version(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int pgl_device_count(void);

/* Context for a population of N neurons observed for nT bins of width dt, with
 * B impulse basis functions of R taps (impulse.py:92-112).  Mirrors
 * Population.__init__/Glm.__init__ (population.py:12-32, glm.py:8-63). */
int pgl_create(int N, int64_t nT, int B, int R, int nlin, double dt, int device,
               pgl_handle* out);
int pgl_destroy(pgl_handle h);
int pgl_set_option(pgl_handle h, int option, int value);

/* Glm.set_data: S.set_value(data["S"]) (glm.py:99-103).  Spike counts (nT,N)
 * row-major.  The f64 form checks that every count is an integer in 0..255
 * (population.py:345-349 caps at 10) and stores uint8. */
int pgl_set_spikes_u8(pgl_handle h, const uint8_t* S);
int pgl_set_spikes_f64(pgl_handle h, const double* S);

/* The interpolated impulse basis `ibasis` (R,B) row-major (impulse.py:92-112, 359-376). */
int pgl_set_basis(pgl_handle h, const double* ibasis);

/* bkgd_model.set_data: stim.set_value(data['fstim']) (bkgd.py:156-157, 342-345).
 * fstim is (nT, Dstim) row-major; Dstim = 0 / NULL = NoStimulus (bkgd.py:29-43). */
int pgl_set_stim_features(pgl_handle h, const double* fstim, int Dstim);

/* BasisStimulus / SpatiotemporalStimulus.preprocess_data on the device (bkgd.py:122-154,
 * 303-340; basis.py:201-273): linear interpolation of the raw stimulus (Tstim, D) from the dt_stim
 * grid to the dt grid (np.interp, clamped at the ends), projection on the spatial basis
 * basis_x (D,Bx) (NULL = identity), strictly causal convolution with every temporal basis
 * basis_t (Rt,Bt).  The features stay on the device as the Dstim = Bx*Bt extra columns of F:
 * layout 0: column bt*Bx+bx (SpatiotemporalStimulus, bkgd.py:337-340), layout 1: column bx*Bt+bt
 * (BasisStimulus d*B+b, bkgd.py:148-152).  pgl_get_stim_features copies them out (nT, Dstim). */
int pgl_set_stimulus(pgl_handle h, const double* stim, int64_t Tstim, int D, double dt_stim,
                     const double* basis_x, int Bx, const double* basis_t, int Rt, int Bt,
                     int layout);
int pgl_get_stim_features(pgl_handle h, double* fstim_out);

/* SpatiotemporalStimulus with a WIDE stimulus (bkgd.py:172-345): the rank-1 structure
 * w_stim = vec(w_t (x) w_x) (bkgd.py:214-220) is kept on the device instead of materialising the
 * (nT, Bt*Bx) feature matrix of pgl_set_stimulus (7.4 GB at D = 1024, T = 300 s):
 *   I_stim[:,n] = causal conv( np.interp( (stim . basis_x) . w_x[n] ), basis_t . w_t[n] )
 * -- one GEMM at the stimulus frame rate plus a 1-D convolution per neuron; gradients by the transposed
 * operations.  After this call a theta row is [bias, w_t(Bt), w_x(Bx), w_imp(N*B)] (the reference's own
 * packing order of 'bkgd': 'w_t' < 'w_x', packvec.py:22) with P = 1 + Bt + Bx + N*B, and gradients come
 * back in that layout (no host chain rule).  basis_x (D,Bx) row-major or NULL = identity.
 * When dt_stim is an integer multiple q of dt, ceil(Rt / q) + 2 <= 8 and Bt <= 4 (the reference's frames of 100
 * bins with Rt = 300 qualify) the evaluation runs at the FRAME rate: the interpolated projection is piecewise
 * linear over q-bin frames, so the Rt taps of a bin collapse to <= 8 frame values through a coefficient table
 * built here (k_sepf_fwd / k_sepf_bwd), and the impulse columns run on resident feature tiles -- for neuron ranges
 * and neuron lists of up to 128 neurons (up to 64 neurons, <= 3 temporal bases, <= 5 frame values and q >= 16 the
 * stimulus current is five more k-steps of the fused kernel's forward contraction instead of a kernel of its own);
 * other ratios and populations of more than 128 neurons keep the tap-rate kernels (same results to 1e-12;
 * pgl_info[12] tells which). */
int pgl_set_stimulus_separable(pgl_handle h, const double* stim, int64_t Tstim, int D, double dt_stim,
                               const double* basis_x, int Bx, const double* basis_t, int Rt, int Bt);

/* Restrict pgl_ll_grad to the bins [t_lo, t_hi) (t_lo a multiple of 16): ll and gradient
 * become the partial sums over that range, while features still see the spikes before t_lo.
 * The likelihood is a sum over data segments (population.py:41-43), so a time range per GPU
 * plus an all-reduce of (ll, grad) shards one evaluation over GPUs.  Default: [0, nT). */
int pgl_set_time_range(pgl_handle h, int64_t t_lo, int64_t t_hi);

/* seval(glm.ll) and seval(g_glm_ll) for every post-synaptic neuron n in
 * [n_lo, n_hi) in one fused pass (glm.py:39-52; coord_descent.py:27-30, 52-57, 73-78;
 * population.py:71-86).  theta is ((n_hi-n_lo), P), Weff is (N,N).
 * ll_out[(n_hi-n_lo)], grad_out[(n_hi-n_lo)*P] (grad_out may be NULL: ll only). */
int pgl_ll_grad(pgl_handle h, int n_lo, int n_hi, const double* theta,
                const double* Weff, double* ll_out, double* grad_out);

/* Same with device pointers, asynchronous on the handle's stream; pair with pgl_sync. */
int pgl_ll_grad_dev(pgl_handle h, int n_lo, int n_hi, const double* d_theta,
                    const double* d_Weff, double* d_ll, double* d_grad);
/* The same for an arbitrary list of post-synaptic neurons: d_idx[j] (device, int32, distinct) is the
 * neuron of row j of d_theta / d_ll / d_grad.  Lets a lock-step optimiser evaluate only the neurons
 * whose line search is still running (the reference fits one neuron per call, coord_descent.py:161-204). */
int pgl_ll_grad_list_dev(pgl_handle h, const int* d_idx, int count, const double* d_theta,
                         const double* d_Weff, double* d_ll, double* d_grad);
int pgl_sync(pgl_handle h);
/* The lock-step optimiser (inference/batched_bfgs.py) as row kernels on the handle's stream.  The reference calls
 * scipy.optimize.minimize(method="bfgs") neuron by neuron (coord_descent.py:161-204); these run the same algorithm --
 * BFGS from H = I, More'-Thuente strong-Wolfe line search with scipy's constants and first trial step
 * (csrc/pglm_linesearch.h), stop on max|g| <= gtol or maxiter iterations -- for all M neurons of a shard at once, one
 * fused ll+grad launch per trial step.  All optimiser state of a shard of M neurons with P parameters lives in ONE
 * device block of pgl_bfgs_state_doubles(M, P) doubles (flags and counters stored as doubles), in this order:
 *   (M,P) each: X, g, p, H g, s, y, t = H g_new, Xb, gb (best trial of the running search);
 *   (M,P,3) each: U, V (pending H += U V^T);
 *   (M) each: f, fprev, alpha, slope, rho, hscale, iters, restarts, active, frozen, acc, upd, stall, ident, pend, fb,
 *             nfev, hk (updates in the history);  then the line-search state, (18, M).
 * The dense inverse Hessians d_H (M, P, ld), ld even and >= P, are the caller's buffer (uninitialised is fine).
 *   init:       X, f, g of every row in place -> steepest-descent start, first trial step min(1, 1.01/|g|)
 *   trial:      Xt[j] = X[r] + alpha[r] p[r], r = d_rows[j] (NULL: j), j < L
 *   objective:  rows of d_Xt are theta rows [bias, w_stim, w_ir]; in place ll -> f = -(ll + log prior),
 *               grad -> g = -(grad + prior gradient) with fit_glm's NaN rules (coord_descent.py:170-182);
 *               priors: bias.py:33, bkgd.py:76 (stim_sigma), priors.py:139 (kind 0) / 202 (kind 1: group lasso)
 *               (rows in another packing: the caller supplies f and g itself)
 *   linesearch: one step of every listed row's search: next trial step, or the row takes the step (acc = 1), or the
 *               search is stuck: best sufficient-decrease point if any, else stall = 1 (scipy stops with
 *               "precision loss" there); at most max_trials steps per search (scipy: 100)
 *   hmul:       rows with acc = 1: H += U V^T (pending update) and t = H g in one pass over H
 *   update:     U, V, H_new g, next direction and first step, restart (once) / freeze of stalled rows, convergence;
 *               init_scaling != 0: H <- (s.y / y.y) I before the first update after a (re)start (not scipy's). */
long long pgl_bfgs_state_doubles(int M, int P);
int pgl_bfgs_init_dev(pgl_handle h, double* d_state, int M, int P, double gtol);
int pgl_bfgs_trial_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, double* d_Xt);
int pgl_bfgs_objective_dev(pgl_handle h, int L, int P, const double* d_Xt, double* d_ll_f, double* d_grad_g,
                           int prior_kind, double mu_b, double sg_b, double stim_sigma, double mu, double sigma,
                           double lam);
int pgl_bfgs_linesearch_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, const double* d_Xt,
                            const double* d_f, const double* d_g, int max_trials);
int pgl_bfgs_hmul_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, double* d_H, int ld);
/* hmul with the inverse Hessians kept implicit: H = hscale I + sum_j U_j V_j^T over the update history that
 * pgl_bfgs_update_dev appends to -- d_hist [M][Kmax][2][P] (s_j, H y_j), d_coef [M][Kmax][2] -- Kmax >= maxiter;
 * d_ab: scratch [M][Kmax][2].  Reads 4 hk P numbers per row instead of 2 P^2: the cheaper form while hk <= P / 2,
 * and no P^2 memory. */
int pgl_bfgs_hmul_hist_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, const double* d_hist,
                           const double* d_coef, int Kmax, double* d_ab);
/* d_hist / d_coef NULL: dense form (pgl_bfgs_hmul_dev) */
int pgl_bfgs_update_dev(pgl_handle h, double* d_state, int M, int P, double gtol, int maxiter, int init_scaling, double* d_hist,
                        double* d_coef, int Kmax);
/* Everything an evaluation of the L listed rows is followed by, as ONE call (and, while the update history is short, ONE
 * row kernel -- one workgroup per row -- instead of objective | linesearch | hmul | update | trial): priors and NaN rules on
 * (d_ll_f, d_grad_g) in place (prior_kind >= 0; < 0: they already hold f and g), the line-search step, t = H g for the rows
 * that took a step (update history d_hist / d_coef / d_ab as for pgl_bfgs_hmul_hist_dev, or dense d_H / ld as for
 * pgl_bfgs_hmul_dev -- exactly one of the two), the update, and the NEXT trial point of every listed row:
 *   d_Xt_next[d_pos_next[r]] = X[r] + alpha[r] p[r]   (d_pos_next (M) int32: position of row r in the next launch's list, < 0
 *   = not listed any more; NULL = same positions as this launch; d_Xt_next NULL = no trial points).
 * flags_out (NULL or M doubles of PINNED host memory): active flag of every listed row, written by the kernel itself -- the
 * driver polls it behind an event, no copy kernel.  hk_bound: upper bound on the updates in any row's history (launches so
 * far), selects the one-kernel or the split form (PGL_OPT_BFGS_MERGE).  coord_descent.py:161-204 is the unit replaced. */
int pgl_bfgs_step_dev(pgl_handle h, double* d_state, int M, int P, const int* d_rows, int L, const double* d_Xt, double* d_ll_f,
                      double* d_grad_g, int prior_kind, double mu_b, double sg_b, double stim_sigma, double mu, double sigma,
                      double lam, int max_trials, double gtol, int maxiter, int init_scaling, double* d_hist, double* d_coef,
                      int Kmax, double* d_ab, int hk_bound, double* d_H, int ld, const int* d_pos_next, double* d_Xt_next,
                      double* flags_out);

/* convolve_with_basis(S, ibasis) (basis.py:201-236 via impulse.py:114-130):
 * fS_out (nT,N,B) row-major, float64. */
int pgl_features(pgl_handle h, double* fS_out);

/* seval(imp_model.I_imp) (impulse.py:58 / 308; gibbs.py:812-833):
 * I_imp_out (nT,N) row-major for impulse weights w (N,B). */
int pgl_impulse_currents(pgl_handle h, const double* w, double* I_imp_out);

/* Glm.get_state lam / I_net / I_bkgd for neuron n (glm.py:75-91, population.py:88-120).
 * theta_n is one row (P); Weff_col is column n of Weff (N).  Outputs (nT) or NULL. */
int pgl_state(pgl_handle h, int n, const double* theta_n, const double* Weff_col,
              double* lam_out, double* I_net_out, double* I_stim_out);

/* CollapsedGibbsNetworkColumnUpdate._glm_ll (gibbs.py:910-937): for each w[k],
 * ll_k = sum_t(-dt*lam + log(lam)*S[t,n_post]), lam = nlin(I_bias + I_stim[t] +
 * I_other[t] + w[k]*I_col[t]).  I_stim may be NULL.  Host arrays of nT. */
int pgl_ll_from_current(pgl_handle h, int n_post, double I_bias, const double* I_stim,
                        const double* I_other, const double* I_col, const double* w,
                        int K, double* ll_out);

/* Device-resident form of _precompute_vars + _precompute_other_current + _glm_ll
 * (gibbs.py:812-864, 910-937).  prepare: computes I_imp (all presynaptic columns)
 * and the total I_net for neuron n_post once.  ll: for presynaptic n_pre, removes
 * the current contribution aw_cur*I_imp[:,n_pre] (rank-1 downdate instead of the
 * reference's full gemv per pair) and evaluates ll at the K candidate weights. */
int pgl_gibbs_prepare(pgl_handle h, int n_post, const double* theta_n,
                      const double* Weff_col);
int pgl_gibbs_ll(pgl_handle h, int n_pre, double aw_cur, const double* w, int K,
                 double* ll_out);
/* After A[n_pre,n_post]*W[n_pre,n_post] changed by `delta` (gibbs.py:1044-1066 writes the
 * new sample into the state dict): I_net += delta * I_imp[:,n_pre] on the device. */
int pgl_gibbs_update(pgl_handle h, int n_pre, double delta);

/* The same collapsed-Gibbs inner loop for MANY columns per launch.  Given the rest of the state the
 * columns (A[:,n], W[:,n]) are conditionally independent -- the reference maps them over its engines
 * (parallel_gibbs.py:162-165, concatenate_parallel_updates 24-37) -- so one call serves one
 * (n_pre, n_post) pair of every listed column.
 *   prepare_all: theta (N,P) flat feature weights of all neurons, Weff (N,N); computes the total
 *     current I_stim + I_net of every post-synaptic neuron once (forward-only MFMA pass,
 *     gibbs.py:812-833 for all n_post) and keeps it, with theta, on the device.  Honours
 *     pgl_set_time_range (ll sums then run over [t_lo, t_hi)).
 *   ll_cols: for column c: n_post[c], n_pre[c], aw_cur[c] = current A*W of the pair, w[c*K .. c*K+K)
 *     candidate weights (K <= 16, e.g. the 10 Gauss-Hermite nodes + w = 0, gibbs.py:1002-1032);
 *     ll_out[c*K + k] as pgl_gibbs_ll.  The impulse weights of the pair are theta[n_post][1+Dstim+n_pre*B ..].
 *     explinear: f64 sums; the log1p(exp(-|x|)) term of bins with 12 <= |x| < 700 comes from the single-precision
 *     hardware exp (absolute error <= 6e-12 per bin; see PGL_OPT_GIBBS_KERNEL for the all-f64 kernel).
 *   update_cols: after A*W of pair c changed by delta[c]: I_net[:, n_post[c]] += delta[c]*I_imp (n_post distinct).
 *   currents: copy out bias-free total current I_stim + I_net of one neuron over the prepared range. */
int pgl_gibbs_prepare_all(pgl_handle h, const double* theta, const double* Weff);
int pgl_gibbs_ll_cols(pgl_handle h, int ncols, const int* n_post, const int* n_pre,
                      const double* aw_cur, const double* w, int K, double* ll_out);
int pgl_gibbs_update_cols(pgl_handle h, int ncols, const int* n_post, const int* n_pre,
                          const double* delta);
int pgl_gibbs_currents(pgl_handle h, int n_post, double* x_out);

/* Spike-triggered average, pyglm/utils/sta.py:6-85 (used by smart_init.py:28-98 and 100-158):
 *   A[i,l,d] = sum_t S[t,n_i] * istim[t-l,d] / sum_t S[t,n_i],  l = 0..L-1, terms with t-l < 0 dropped,
 * istim = np.interp of stim (Tstim,D) to the bin grid, divided by dt_stim/dt (sta.py:27-41).
 *   neurons: n_sel indices (NULL = all N);  A_out host (n_sel, L, D).  A silent neuron gives NaN
 *   like the reference's 0/0.  Uses the spike data already resident on the handle. */
int pgl_sta(pgl_handle h, const double* stim, int64_t Tstim, int D, double dt_stim, int L,
            const int* neurons, int n_sel, double* A_out);

/* Leading singular pair (u_0, sigma_0, v_0) of each of n (L x D) row-major matrices -- what initialize_stim_with_sta keeps of
 * np.linalg.svd(STA) (smart_init.py:66-72).  Host arrays: A (n, L, D) in; U (n, L), sigma (n), V (n, D) out; the
 * component of u_0 of largest magnitude is positive.  Device work with the library's own kernels (Gram matrix of the
 * smaller side, repeated squaring, two alternating steps on A).  A == NULL: the averages a preceding pgl_sta call with
 * A_out == NULL has left on the device (same n, L, D; consumed by this call) -- the initialisation of a wide stimulus never
 * moves its 157 MB of averages over PCIe. */
int pgl_leading_singular_pairs(pgl_handle h, const double* A, int n, int L, int D, double* U, double* sigma, double* V);

/* Population.simulate (population.py:233-389), native host implementation (no GPU needed):
 * integrate-and-fire thinning of the conditional intensity.  Per bin t: lam = nlin(X[t,:]),
 * acc += lam*dt, a neuron spikes while acc > thr (thr ~ Exp(1), redrawn after each spike,
 * population.py:321-360); every spike of n_pre adds AW[n_pre,:,n_post] to X[t+1 : t+R+1, n_post]
 * (351-353); at most 10 spikes per bin (345-349: the round that would exceed the cap is dropped).
 *   X     (nT,N) in/out: on entry bias + stimulus current (population.py:252-268), on exit the
 *         total current; S (nT,N) out, float64 counts.
 *   AW    (N, R, N) = A[n_pre,n_post]*W[n_pre,n_post]*impulse[n_pre->n_post][tau], layout
 *         [n_pre][tau][n_post].
 *   uniforms: optional stream of U(0,1) numbers consumed in the reference's draw order (N for the
 *         initial thresholds, then one per spiking neuron per round); when exhausted (or NULL) a
 *         splitmix64 generator seeded with `seed` continues.  n_exceptions_out may be NULL. */
int pgl_simulate(int N, int64_t nT, int R, int nlin, double dt, double* X, const double* AW,
                 const double* uniforms, int64_t n_uniforms, uint64_t seed, double* S,
                 int64_t* n_exceptions_out);

/* Timing of the most recent pgl_ll_grad[_dev] call, measured with HIP events on the
 * handle's stream: ms of the fused kernel alone and of the whole call (prep +
 * fused + finalize).  For the _dev form call after pgl_sync. */
int pgl_last_timing(pgl_handle h, double* fused_ms, double* total_ms);

/* Mean of the same two figures over the pgl_ll_grad[_dev] calls since the last reset (at most the
 * 256 most recent ones; each call records its own HIP event set, so a caller can queue many
 * evaluations without a host synchronisation in between).  Synchronises the stream; reset != 0
 * starts a new window. */
int pgl_timing_summary(pgl_handle h, int reset, int* n_launches, double* mean_fused_ms,
                       double* mean_total_ms);

/* Order all subsequent work of the handle on a caller-owned HIP stream (hipStream_t passed as
 * void*; NULL = back to the handle's own stream).  A caller that runs collectives on its own
 * stream (RCCL through torch.distributed) passes that stream here: evaluation and all-reduce are
 * then ordered by the stream, with no host synchronisation between steps.  The reference has no
 * counterpart (Theano's shared variables are synchronous). */
int pgl_set_stream(pgl_handle h, void* stream);

/* Dev / test: dry run of the kernel dispatch, no device needed -- the names (as in the code object) of the fused kernel
 * instantiations an evaluation of `count` neurons from n_lo of a population of this shape would launch, one per line.
 * stim: 0 none / dense stimulus columns, 1 separable by the tap-rate kernels, 2 separable at the frame rate (stimulus
 * current inside the fused forward where that form exists), 3 at the frame rate through the slab.  path: 0 ll+grad,
 * 1 ll only, 2 the forward launches of pgl_gibbs_prepare_all.  The reference has no counterpart (Theano picks its own
 * C implementations); tests hold every reachable instantiation to zero bytes of scratch. */
int pgl_plan_kernels(int N, int B, int R, int Dstim, long long nT, int stim, int n_lo, int count, int path, int opt_kernel,
                     int opt_f32, char* out, int cap);

/* Launch geometry and algorithmic work of the fused kernel for [n_lo,n_hi):
 * info[0]=blocks, [1]=threads/block, [2]=time chunks, [3]=k-tiles(16 rows),
 * [4]=LDS bytes, [5]=rows per time tile, [6]=algorithmic flops (4*nT*Ktot*npost),
 * [7]=algorithmic bytes, [8]=number of spike events (nonzero bins), [9]=kernel the call would use
 * (1 4-wave, 2 K-split, 3 K-split with f32 features, 4 two-pass, 5 two-pass on resident feature
 * tiles, 6 K-split on resident feature tiles -- incl. the block-ring form k_fused8 for one post tile of a 25..40 k-tile
 * row --, 7 single pass without K split on resident tiles), [10]=bytes of resident feature tiles (0 unless [9] >= 5),
 * [11]=HBM bytes the hot kernels
 * stream per evaluation on top of the algorithmic ones (feature tiles, residual slab), [12]=stimulus path of the
 * call: 0 none / dense feature columns, 1 separable by the tap-rate kernels, 2 separable at the frame rate.
 * (For a separable stimulus [6] counts the impulse contraction only and [7] the projected stimulus at its frame
 * rate.) */
int pgl_info(pgl_handle h, int n_lo, int n_hi, double* info, int n_info);

#ifdef __cplusplus
}
#endif
#endif
