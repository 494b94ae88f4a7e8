"""The set of fused kernel instantiations the dispatcher can reach, from a dry run of make_plan / the launch switches over
a grid of shapes (pgl_plan_kernels: no GPU needed), against the instantiations in the built library:

    python tools/reachable_kernels.py            # summary + instantiations no plan reaches + reachable ones with scratch

`reachable(auto_only)` is what tests/test_capi_symbols.py uses: every instantiation reachable WITHOUT a forcing option
must exist in the library and use no scratch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theano_pyglm_amd import _lib

FUSED = ('k_fused2<', 'k_fused3<', 'k_fused5<', 'k_fused6<', 'k_fused7<', 'k_fused8<', 'k_fused<')


def shapes():
    Ns = list(range(1, 137)) + list(range(144, 521, 8))
    for N in Ns:
        counts = sorted(set(c for c in (N, 1, 15, 16, 17, 32, 33, 48, 49, 64, 65, 80, 100, 128) if c <= N))
        for B in range(1, 9):
            for nT in (16, 64, 600000):          # (one and four 16-bin tiles: the kernels' tile-pair forms need four)
                for stim, Ds in ((0, 0), (0, 2), (0, 9), (0, 200), (1, 3 + 24), (2, 3 + 24), (3, 3 + 24), (2, 4 + 1024)):
                    for count in counts:
                        yield N, B, nT, stim, Ds, count


def reachable_both():
    """({kernel name: example shape} for the automatic dispatch, the same including the forcing options)"""
    auto, forced = {}, {}
    for N, B, nT, stim, Ds, count in shapes():
        for ok in (0, 2, 3, 4, 6, 7):
            for f32 in ((0, 1, 2) if stim == 0 else (0,)):           # (2: f32 resident blocks of the narrow-shard kernel)
                for path in (0, 1, 2):
                    if path == 2 and count != N:
                        continue
                    key = (N, B, nT, stim, Ds, count, path, ok, f32)
                    try:
                        names = _lib.plan_kernels(N, B=B, R=200, Dstim=Ds, nT=nT, stim=stim, count=count, path=path,
                                                  opt_kernel=ok, opt_f32=f32)
                    except _lib.PglError as e:
                        if 'no kernel instantiation' not in str(e):
                            continue                 # no plan for this shape (the evaluation raises the same error)
                        names = ['MISSING: N=%d B=%d nT=%d stim=%d Dstim=%d count=%d path=%d opt_kernel=%d f32=%d' % key]
                    for n in names:
                        forced.setdefault(n, key)
                        if ok == 0:
                            auto.setdefault(n, key)
    return auto, forced


def reachable(auto_only=True):
    return reachable_both()[0 if auto_only else 1]


def built_fused():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import kernel_resources as KR
    res = KR.kernel_resources(_lib.LIB_PATH)
    return dict((KR.short(n), r) for n, r in res.items())


if __name__ == '__main__':
    auto, forced = reachable_both()
    built = built_fused()
    bf = dict((n, r) for n, r in built.items() if n.startswith(FUSED))
    print("fused instantiations built: %d; reachable by the automatic dispatch: %d; reachable with a forcing option: %d"
          % (len(bf), len(auto), len(forced)))
    miss = sorted(n for n in forced if n not in bf)
    print("reachable but NOT built (%d): %s" % (len(miss), miss))
    dead = sorted(n for n in bf if n not in forced)
    print("built but not reachable (%d):" % len(dead))
    for n in dead:
        print("   ", n)
    bad = sorted(n for n in auto if n in bf and bf[n]['scratch'] > 0)
    print("automatically dispatched with scratch (%d): %s" % (len(bad), [(n, bf[n]['scratch'], auto[n]) for n in bad]))
    bad2 = sorted(n for n in forced if n in bf and bf[n]['scratch'] > 0 and n not in auto)
    print("reachable only with a forcing option, with scratch (%d): %s" % (len(bad2), [(n, bf[n]['scratch'], forced[n]) for n in bad2]))
