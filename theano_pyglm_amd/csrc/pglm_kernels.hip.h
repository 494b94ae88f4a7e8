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
//
// The kernels live in one header per family; this file is the translation unit's table of contents.
#include "pglm_common.hip.h"
#include "pglm_fused_stream.hip.h"
#include "pglm_fused_resident.hip.h"
#include "pglm_reduce.hip.h"
#include "pglm_direct.hip.h"
#include "pglm_gibbs.hip.h"
#include "pglm_stim.hip.h"
#include "pglm_bfgs.hip.h"
