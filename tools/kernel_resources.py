"""Register / scratch / LDS use of every kernel in the built library, read from the gfx950 code object's metadata
(llvm-readelf --notes on the device ELF embedded in the .so).  Used by tests/test_capi_symbols.py (no kernel the
dispatcher can reach may use scratch) and as a dev tool:

    python tools/kernel_resources.py [lib.so] [--scratch-only]
"""
import os, re, subprocess, sys, tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def extract_code_object(lib, outdir):
    """The gfx950 device ELF of a hipcc-built shared library (the .hip_fatbin section is a clang offload bundle)."""
    fat = os.path.join(outdir, 'fatbin')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
    co = os.path.join(outdir, 'gfx950.co')
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                           '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return co


def kernel_resources(lib):
    """{demangled kernel name: {'scratch': bytes, 'vgpr': n, 'agpr': n, 'sgpr': n, 'lds': bytes, 'spill_vgpr': n}}"""
    with tempfile.TemporaryDirectory() as d:
        # a bare device ELF (hipcc -save-temps: *-gfx950.out) is read as it is
        co = lib if lib.endswith(('.out', '.co', '.hsaco')) else extract_code_object(lib, d)
        notes = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', co]).decode()
    out = {}
    cur = {}
    for line in notes.splitlines():
        m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == 'agpr_count' and cur:
            pass
        if k in ('agpr_count', 'group_segment_fixed_size', 'private_segment_fixed_size', 'sgpr_count', 'vgpr_count',
                 'vgpr_spill_count', 'sgpr_spill_count', 'name', 'symbol'):
            cur[k] = v
        if k == 'wavefront_size':          # last key of a kernel entry (keys are sorted)
            if 'name' in cur:
                out[cur['name']] = cur
            cur = {}
    names = list(out)
    if names:
        dem = subprocess.check_output(['c++filt'] + names).decode().splitlines()
    else:
        dem = []
    res = {}
    for n, dn in zip(names, dem):
        c = out[n]
        res[dn] = {'scratch': int(c.get('private_segment_fixed_size', 0)), 'vgpr': int(c.get('vgpr_count', 0)),
                   'agpr': int(c.get('agpr_count', 0)), 'sgpr': int(c.get('sgpr_count', 0)),
                   'lds': int(c.get('group_segment_fixed_size', 0)), 'spill_vgpr': int(c.get('vgpr_spill_count', 0)),
                   'spill_sgpr': int(c.get('sgpr_spill_count', 0))}
    return res


def short(name):
    return re.sub(r'\(.*$', '', name).replace('void ', '')


if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    lib = args[0] if args else os.path.join(root, 'theano_pyglm_amd', 'libpyglm_hip.so')
    res = kernel_resources(lib)
    only = '--scratch-only' in sys.argv
    n = 0
    for name in sorted(res, key=short):
        r = res[name]
        if only and r['scratch'] == 0:
            continue
        n += 1
        print("%-64s scratch %4d B  vgpr %3d agpr %3d sgpr %3d lds %6d  spills v%d s%d"
              % (short(name)[:64], r['scratch'], r['vgpr'], r['agpr'], r['sgpr'], r['lds'], r['spill_vgpr'], r['spill_sgpr']))
    print("%d kernels listed of %d" % (n, len(res)))
