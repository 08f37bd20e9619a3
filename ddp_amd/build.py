"""Build libddp_mi355x.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libddp_mi355x.so')
SOURCES = ['ddp_api.hip', 'ddp_gemm.hip', 'ddp_gemm_bf16.hip', 'ddp_kernels.hip', 'ddp_layer_tail.hip']
HEADERS = ['exports.map', 'ddp_internal.h', 'gemm_f32.h', 'gemm_bf16x3.h', 'layer_bf16x3.h', os.path.join('..', '..', 'include', 'ddp_mi355x.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def source_hash():
    """sha256 (16 hex digits) over the kernel sources + the C-ABI header: names the code state a profile was taken from
    (profiles/*_pmc_summary.json carry it; bench.py only quotes PMC traffic measured on the SAME sources)."""
    import hashlib
    h = hashlib.sha256()
    for s in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, s), 'rb') as f:
            h.update(s.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(LIB_DIR, s.replace('.hip', '.o'))
        cmd = [_hipcc()] + FLAGS + ['-x', 'hip', '-c', os.path.join(CSRC, s), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--version-script=' + os.path.join(CSRC, 'exports.map'),
           '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB_PATH)
