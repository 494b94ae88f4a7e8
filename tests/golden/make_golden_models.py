"""
Golden vectors from the reference's own pyglm/models/model_factory.py (pure numpy / scipy; Python-2
syntax converted IN MEMORY by lib2to3 -- relative imports and print statements only -- nothing of its
text is stored).  Run in the build container only:

    python tests/golden/make_golden_models.py     ->  tests/golden/models_golden.json / .npz

  * make_model + stabilize_sparsity (model_factory.py:18-171) for the three templates at N = 4, 32, 128
  * check_stability (:173-185) on seeded (A, W) draws
  * convert_model (:187-268), basis -> dirichlet: the reference needs populations only for
    `eval_state` (impulse responses / target basis) and `glm.imp_model.alpha / B`; it gets duck-typed
    stand-ins whose state comes from THIS repository's host components (the inputs of the fixture).
    Its adjacency thresholding indexes with a float (model_factory.py:251, old-numpy semantics), so the
    call uses a graph without 'rho' (A = ones); the projection, the weights and the Dirichlet
    parameters are what gets pinned.
"""
import copy
import json
import os
import sys
import types
import warnings

import numpy as np

REFROOT = '/root/reference'
REF = REFROOT + '/pyglm/models/model_factory.py'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_reference():
    warnings.simplefilter('ignore')
    np.int = int
    np.float = float
    np.rank = np.ndim
    np.Inf = np.inf                       # alias removed in numpy 2 (model_factory.py:230)
    sys.path.insert(0, REFROOT)
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    tool = RefactoringTool(get_fixers_from_package('lib2to3.fixes'))
    ns = {'__name__': 'pyglm.models.model_factory', '__package__': 'pyglm.models'}
    exec(compile(str(tool.refactor_string(open(REF).read(), REF)), REF, 'exec'), ns)
    return ns


def jsonable(d):
    if isinstance(d, dict):
        return dict((k, jsonable(v)) for k, v in d.items())
    if isinstance(d, (list, tuple)):
        return [jsonable(v) for v in d]
    if isinstance(d, (np.floating, np.integer)):
        return d.item()
    if isinstance(d, np.ndarray):
        return d.tolist()
    return d


def main():
    ref = load_reference()
    here = os.path.dirname(os.path.abspath(__file__))
    out = {'models': {}, 'stability': []}
    for name in ('standard_glm', 'sparse_weighted_model', 'spatiotemporal_glm'):
        for N in (4, 32, 128):
            m = ref['make_model'](name, N=N, dt=0.001)
            ref['stabilize_sparsity'](m)
            out['models']['%s/%d' % (name, N)] = jsonable(m)
    rng = np.random.RandomState(31)
    model = ref['make_model']('sparse_weighted_model', N=5, dt=0.001)
    A_list, W_list = [], []
    for scale in (0.1, 0.4, 0.8, 1.5):
        A = (rng.rand(5, 5) < 0.6).astype(np.int8)
        W = scale * rng.randn(25)
        x = {'net': {'graph': {'A': A}, 'weights': {'W': W}}}
        out['stability'].append(bool(ref['check_stability'](model, x, 5)))
        A_list.append(A)
        W_list.append(W)

    # ---- convert_model (basis -> dirichlet) on inputs from this repository's host components ----
    sys.path.insert(0, ROOT)
    from theano_pyglm_amd.models.model_factory import make_model as our_make_model
    from theano_pyglm_amd.population import Population
    N = 3
    from_model = our_make_model('standard_glm', N=N, dt=0.001)
    to_model = our_make_model('sparse_weighted_model', N=N, dt=0.001)
    from_popn, to_popn = Population(from_model), Population(to_model)
    from_vars = from_popn.sample(np.random.RandomState(7))
    to_vars = to_popn.sample(np.random.RandomState(8))
    impulses = np.array([from_popn.glm.imp_model.impulse(from_vars['glms'][n]['imp']) for n in range(N)])  # [n2][n1][R]
    basis = to_popn.glm.imp_model.ibasis
    fake_from = types.SimpleNamespace(
        N=N, eval_state=lambda v: {'glms': [{'imp': {'impulse': impulses[n]}} for n in range(N)]})
    fake_to = types.SimpleNamespace(
        N=N, eval_state=lambda v: {'glms': [{'imp': {'basis': basis}} for n in range(N)]},
        glm=types.SimpleNamespace(imp_model=types.SimpleNamespace(alpha=to_popn.glm.imp_model.alpha,
                                                                  B=to_popn.glm.imp_model.B)))
    to_model_ref = copy.deepcopy(to_model)
    to_model_ref['network']['graph'].pop('rho', None)          # float index at model_factory.py:251
    to_model_ref['latent'] = {}
    conv = ref['convert_model'](fake_from, copy.deepcopy(from_model), copy.deepcopy(from_vars), fake_to,
                                to_model_ref, copy.deepcopy(to_vars))
    g = np.array([[conv['glms'][n2]['imp']['g_%d' % n1] for n1 in range(N)] for n2 in range(N)])
    np.savez(os.path.join(here, 'models_golden.npz'),
             stab_A=np.array(A_list), stab_W=np.array(W_list),
             conv_impulses=impulses, conv_basis=basis, conv_g=g,
             conv_W=np.asarray(conv['net']['weights']['W']), conv_A=np.asarray(conv['net']['graph']['A']),
             conv_bias=np.array([conv['glms'][n]['bias']['bias'] for n in range(N)]),
             from_bias=np.array([from_vars['glms'][n]['bias']['bias'] for n in range(N)]),
             from_w_ir=np.array([from_vars['glms'][n]['imp']['w_ir'] for n in range(N)]))
    with open(os.path.join(here, 'models_golden.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote models_golden.json / .npz; stability:", out['stability'])


if __name__ == '__main__':
    main()
