"""
Golden vectors for the packed parameter layout from the reference's own
pyglm/utils/packvec.py: `unpackdict`, `get_vars` and `set_vars` run unchanged (Python-2 `print`
converted in memory by lib2to3); `packdict` itself cannot run under numpy 2 (`val == []` on arrays,
packvec.py:27) -- its visiting order is the same sorted-key recursion that `unpackdict` applies, so the
inverse pins the layout.  Run in the build container only:

    python tests/golden/make_golden_packvec.py     ->  tests/golden/packvec_golden.json

For every model the fixture holds the per-neuron shapes dict of the differentiable GLM variables and,
for the vector 0, 1, 2, ..., the nested dict the reference's unpackdict returns (i.e. which entries of
the packed vector belong to which variable).
"""
import json
import os
import sys
import warnings

import numpy as np

REF = '/root/reference/pyglm/utils/packvec.py'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_reference():
    warnings.simplefilter('ignore')
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    tool = RefactoringTool(get_fixers_from_package('lib2to3.fixes'))
    src = open(REF).read()
    if not src.endswith('\n'):
        src += '\n'
    ns = {}
    exec(compile(str(tool.refactor_string(src, REF)), REF, 'exec'), ns)
    return ns


def to_lists(d):
    return dict((k, to_lists(v) if isinstance(v, dict) else np.asarray(v).tolist()) for k, v in d.items())


def shapes_to_lists(d):
    return dict((k, shapes_to_lists(v) if isinstance(v, dict) else list(v)) for k, v in d.items())


def main():
    ref = load_reference()
    sys.path.insert(0, ROOT)
    from theano_pyglm_amd.models.model_factory import make_model
    from theano_pyglm_amd.population import Population
    from theano_pyglm_amd.utils import packvec as PV
    out = {}
    for name, N in (('standard_glm', 3), ('sparse_weighted_model', 3), ('spatiotemporal_glm', 4)):
        popn = Population(make_model(name, N=N, dt=0.001))
        x = popn.sample(np.random.RandomState(1))
        syms = popn.glm_syms()
        xv = ref['get_vars'](syms, x['glms'][1])                 # reference get_vars on our state dict
        _, shapes = PV.packdict(xv)                              # shapes only; the ORDER comes from the reference below
        size = int(sum(np.prod(s) for _, s in PV._walk(shapes)))
        unpacked = ref['unpackdict'](np.arange(float(size)), shapes)
        # set_vars: write the unpacked values into a copy of the neuron's variables
        import copy
        target = copy.deepcopy(x['glms'][1])
        ref['set_vars'](syms, target, unpacked)
        out[name] = {'N': N, 'size': size, 'shapes': shapes_to_lists(shapes), 'unpacked': to_lists(unpacked),
                     'after_set_vars': to_lists(dict((k, v) for k, v in target.items() if isinstance(v, dict)))}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, 'packvec_golden.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote packvec_golden.json:", dict((k, v['size']) for k, v in out.items()))


if __name__ == '__main__':
    main()
