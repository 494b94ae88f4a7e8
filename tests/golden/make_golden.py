"""
Generate golden vectors from the importable part of the reference
(/root/reference/pyglm/utils/basis.py + the model-template dicts) -- SURVEY.md §8(c).

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs small .npz fixtures next to this script.  The fixtures hold inputs and
the reference's outputs (data only; no reference source text).
"""
import os
import sys
import numpy as np

REF = '/root/reference'


def main():
    # alias shims needed by the py2-era reference under numpy 2.x (basis.py:84, fftconv.py:78)
    np.int = int
    np.float = float
    np.rank = np.ndim
    sys.path.insert(0, REF)
    from pyglm.utils import basis as rb
    from pyglm.models.standard_glm import StandardGlm
    from pyglm.models.sparse_weighted_model import SparseWeightedModel
    from pyglm.models.spatiotemporal_glm import SpatiotemporalGlm

    here = os.path.dirname(os.path.abspath(__file__))
    out = {}

    # --- A1: raw 100-point bases of the three templates -------------------
    out['std_imp_basis'] = rb.create_basis(StandardGlm['impulse']['basis'])          # cosine, orth
    out['std_bkgd_basis'] = rb.create_basis(StandardGlm['bkgd']['basis'])            # cosine, orth
    out['swm_imp_basis'] = rb.create_basis(SparseWeightedModel['impulse']['basis'])  # cosine, norm
    out['st_imp_basis'] = rb.create_basis(SpatiotemporalGlm['impulse']['basis'])     # cosine, norm
    out['st_temporal_basis'] = rb.create_basis(SpatiotemporalGlm['bkgd']['temporal_basis'])
    out['st_spatial_basis'] = rb.create_basis(SpatiotemporalGlm['bkgd']['spatial_basis'])
    # un-orthogonalised, un-normalised variant (pure formula, platform independent)
    p = dict(StandardGlm['impulse']['basis']); p['orth'] = False
    out['cos5_raw_basis'] = rb.create_basis(p)

    # --- A2: convolve_with_basis on seeded spike counts ----------------------
    rng = np.random.default_rng(20260101)
    T, N = 700, 4
    S = rng.poisson(0.04, size=(T, N)).astype(float)
    S[5, 0] = 3.0                                  # multi-spike bin
    S[T - 1, 1] = 1.0                              # spike in the last bin (contributes nothing)
    # interpolation of the basis to R taps as in impulse.py:92-103 (restated; linspace(0,1,R))
    R = 200
    L, B = out['std_imp_basis'].shape
    ib = np.zeros((R, B))
    for b in range(B):
        ib[:, b] = np.interp(np.linspace(0, 1, R), np.linspace(0, 1, L), out['std_imp_basis'][:, b])
    out['conv_S'] = S
    out['conv_ibasis'] = ib
    out['conv_fS'] = rb.convolve_with_basis(S, ib)

    # short-signal edge cases: T < R and T == 1
    S2 = rng.poisson(0.1, size=(37, 2)).astype(float)
    out['conv_short_S'] = S2
    out['conv_short_fS'] = rb.convolve_with_basis(S2, ib)
    S3 = np.array([[2.0, 0.0, 1.0]])
    out['conv_one_S'] = S3
    out['conv_one_fS'] = rb.convolve_with_basis(S3, ib)

    # --- A3: convolve_with_low_rank_2d_basis ---------------------------------
    T3, D = 500, 3
    stim = rng.standard_normal((T3, D))
    Rt = 300
    Lt, Bt = out['st_temporal_basis'].shape
    ibt = np.zeros((Rt, Bt))
    for b in range(Bt):
        ibt[:, b] = np.interp(np.linspace(0, 1, Rt), np.linspace(0, 1, Lt), out['st_temporal_basis'][:, b])
    ibt = ibt / np.tile(np.sum(ibt, 0), [Rt, 1])
    ibx = out['st_spatial_basis']
    out['lr2d_stim'] = stim
    out['lr2d_ibasis_t'] = ibt
    out['lr2d_ibasis_x'] = ibx
    out['lr2d_fstim'] = rb.convolve_with_low_rank_2d_basis(stim, ibx, ibt)

    # --- reference self-check 2 (basis.py:438-451): one-bin causal shift -----
    st = rng.standard_normal((60, 1))
    one = np.array([[1.0]])
    out['shift_stim'] = st
    out['shift_fstim'] = rb.convolve_with_basis(st, one)

    # --- project_onto_basis (basis.py:416-436), used by the STA initialisation ------------
    rng2 = np.random.default_rng(20260202)
    f = rng2.standard_normal(Rt)
    out['proj_f'] = f
    out['proj_beta'] = rb.project_onto_basis(f, ibt)
    out['proj_beta_ridge'] = rb.project_onto_basis(f, ibt, lam=0.5)

    np.savez_compressed(os.path.join(here, 'basis_golden.npz'), **out)

    # --- the three template dicts (hyper-parameters only: data, not code) -------------
    import json
    with open(os.path.join(here, 'templates_golden.json'), 'w') as f:
        json.dump({'standard_glm': StandardGlm, 'sparse_weighted_model': SparseWeightedModel,
                   'spatiotemporal_glm': SpatiotemporalGlm}, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, v.shape)


if __name__ == '__main__':
    main()
