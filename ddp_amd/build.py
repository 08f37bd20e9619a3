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


STAMP_PATH = os.path.join(LIB_DIR, '.source_sha')


def built_hash():
    """source_hash() of the sources the .so in lib/ was built from ('' when there is no stamp)."""
    try:
        with open(STAMP_PATH) as f:
            return f.read().strip()
    except OSError:
        return ''


def needs_build():
    # by CONTENT, not by mtime: after a fresh checkout (or a gpurun snapshot) mtimes are arbitrary, and the .so is git-ignored
    return not os.path.exists(LIB_PATH) or built_hash() != source_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    sha = source_hash()                  # hashed BEFORE compiling: an edit during the build leaves a stale stamp, i.e. a rebuild
    if os.path.exists(STAMP_PATH):
        os.remove(STAMP_PATH)
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
    with open(STAMP_PATH, 'w') as f:
        f.write(sha + '\n')
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB_PATH)
