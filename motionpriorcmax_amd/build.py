"""Compile libmpcmax.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmpcmax.so')
SOURCES = ['api.hip', 'contrast.hip', 'events.hip', 'knn.hip', 'knn_strip.hip', 'voxel.hip', 'ingest.hip', 'flow.hip', 'curves.hip', 'tiles.hip']
FLAGS = ['-shared', '-fPIC', '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-std=c++17',
         '-Wall', '-Wno-unused-function']


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and the C header: the identity of the library for the profiles that are
    committed beside it.  (The hash of libmpcmax.so itself changes with every build: hipcc puts a random unit id into every
    translation unit -- a profile taken before a rebuild of the same sources would count as stale.)"""
    import hashlib
    h = hashlib.sha256()
    files = []
    for root, _, names in os.walk(CSRC):
        files += [os.path.join(root, n) for n in names if n.endswith(('.hip', '.h'))]
    files.append(os.path.join(HERE, '..', 'include', 'mpcmax.h'))
    for f in sorted(files):
        h.update(os.path.relpath(f, HERE).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, '..', 'include', 'mpcmax.h')]
    for root, _, names in os.walk(CSRC):          # (csrc/diag/*.h too)
        deps += [os.path.join(root, n) for n in names]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, out=None):
    """hipcc every source to an object (in parallel: one process per file), then link.  `out`: another file name for the
    library (diagnostic / bounds-checked builds loaded through MPC_AB_LIB); the product build is LIB."""
    out = os.path.abspath(out) if out else LIB          # (the compiler runs in csrc/)
    if not force and out == LIB and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # MPC_EXTRA_HIPCC_FLAGS: tuning sweeps (-DEV_LUT_INFLIGHT=4 ...); never set for a product build
    extra = os.environ.get('MPC_EXTRA_HIPCC_FLAGS', '').split()
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory(prefix='mpcbuild_') as td:
        def compile_one(src):
            obj = os.path.join(td, src.replace('.hip', '.o'))
            cmd = [hipcc, '-c'] + [f for f in FLAGS if f != '-shared'] + extra + ['-o', obj, os.path.join(CSRC, src)]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True, cwd=CSRC)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        cmd = [hipcc, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', out] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    return out


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
