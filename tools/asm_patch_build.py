"""Experimental build: compile pglm_capi.hip with -save-temps, rewrite the device assembly (VOP2 v_cndmask_b32 ->
VOP3 encoding: tools/ubench/valu_rates_ubench.hip measures the e32 form at 16-20 cycles per instruction and SIMD when
several SIMDs issue it, 4.35 for e64), reassemble and relink.  Usage: asm_patch_build.py <out.so> [extra hipcc flags]"""
import os, re, shlex, subprocess, sys
out = os.path.abspath(sys.argv[1]); extra = sys.argv[2:]
tmp = out + '.tmp'; os.makedirs(tmp, exist_ok=True)
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'theano_pyglm_amd', 'csrc')
base = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-save-temps=obj'] + extra + ['pglm_capi.hip', '-o', os.path.join(tmp, 'lib.so')]
subprocess.run(base, cwd=src, check=True, stderr=subprocess.DEVNULL)
cmds = subprocess.run(base + ['-###'], cwd=src, capture_output=True, text=True).stderr.splitlines()
cmds = [shlex.split(c) for c in cmds if c.startswith(' "')]
dev_s = os.path.join(tmp, 'pglm_capi-hip-amdgcn-amd-amdhsa-gfx950.s')
host_s = os.path.join(tmp, 'pglm_capi-host-x86_64-unknown-linux-gnu.s')
txt = open(dev_s).read()
n0 = txt.count('v_cndmask_b32_e32')
txt = re.sub(r'v_cndmask_b32_e32 (v\d+), ([^,]+), ([^,]+), vcc\b', r'v_cndmask_b32_e64 \1, \2, \3, vcc', txt)
print('rewrote', n0 - txt.count('v_cndmask_b32_e32'), 'of', n0, 'VOP2 v_cndmask_b32')
open(dev_s, 'w').write(txt)
def run(pred):
    for c in cmds:
        if pred(c):
            subprocess.run(c, cwd=src, check=True); return
    raise SystemExit('command not found')
run(lambda c: '-cc1as' in c and 'amdgcn-amd-amdhsa' in c)
run(lambda c: c[0].endswith('lld') and 'elf64_amdgpu' in c)
run(lambda c: 'clang-offload-bundler' in c[0])
fb = os.path.join(tmp, 'pglm_capi.hip-hip-amdgcn-amd-amdhsa.hipfb')
lines = open(host_s).read().split('\n')
for i, l in enumerate(lines):
    if l.startswith('\t.asciz\t"__CLANG_OFFLOAD_BUNDLE__'):
        lines[i] = '\t.incbin\t"%s"' % fb
        m = re.match(r'\t\.size\t(\S+), \d+', lines[i + 1]); assert m
        lines[i + 1] = '\t.size\t%s, %d' % (m.group(1), os.path.getsize(fb))
        break
else:
    raise SystemExit('fatbin blob not found in the host assembly')
open(host_s, 'w').write('\n'.join(lines))
run(lambda c: '-cc1as' in c and 'x86_64-unknown-linux-gnu' in c)
for c in cmds:
    if c[0].endswith('ld.lld'):
        c[c.index('-o') + 1] = out
        subprocess.run(c, cwd=src, check=True)
print('built', out)
