"""
Golden vectors for the categorical draw of the Gibbs updates, from the reference's own
pyglm/inference/log_sum_exp.py (pure numpy, Python-2 syntax: its two `print` statements are converted
in memory by lib2to3; nothing of its text is stored).  Run in the build container only:

    python tests/golden/make_golden_lse.py     ->  tests/golden/lse_golden.npz

The reference draws u from the GLOBAL numpy stream (log_sum_exp.py:26): every case seeds it, the fixture
keeps (lnp, seed, choice).
"""
import os
import warnings
import numpy as np

REF = '/root/reference/pyglm/inference/log_sum_exp.py'


def load_reference():
    warnings.simplefilter('ignore')
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    tool = RefactoringTool(get_fixers_from_package('lib2to3.fixes'))
    src = open(REF).read()
    ns = {}
    exec(compile(str(tool.refactor_string(src, REF)), REF, 'exec'), ns)
    return ns['log_sum_exp_sample']


def main():
    ref = load_reference()
    rng = np.random.default_rng(20260102)
    lnps, seeds, choices, lens = [], [], [], []
    for case in range(60):
        n = int(rng.integers(2, 9))
        lnp = rng.normal(0.0, 3.0, size=n)
        if case % 7 == 0:
            lnp[int(rng.integers(0, n))] = -np.inf            # impossible entry
        if case % 11 == 0:
            lnp = lnp - 800.0                                 # far below the exp range without the max shift
        seed = 1000 + case
        np.random.seed(seed)
        c = ref(lnp.copy())
        row = np.full(8, np.nan)
        row[:n] = lnp
        lnps.append(row); seeds.append(seed); choices.append(c); lens.append(n)
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez(os.path.join(here, 'lse_golden.npz'), lnp=np.array(lnps), n=np.array(lens), seed=np.array(seeds),
             choice=np.array(choices))
    print("wrote lse_golden.npz:", len(choices), "cases; choices", np.bincount(choices))


if __name__ == '__main__':
    main()
