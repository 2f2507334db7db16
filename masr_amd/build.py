"""Build libmasr_hip.so for gfx950 with hipcc (in-tree, no JIT cache)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, 'csrc')
LIB_DIR = os.path.join(ROOT, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libmasr_hip.so')
SOURCES = ['gemm_f32.hip', 'ffn_fused.hip', 'ffn_pc.hip', 'rowgemm.hip', 'rowgemm_small.hip', 'elementwise.hip', 'attention.hip', 'lstm.hip', 'beam_gpu.hip', 'fbank.hip', 'engine.hip', 'beam_search.cpp']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(ROOT), 'include', 'masr_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> masr_amd/lib/libmasr_hip.so"""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + '.o')
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-c',
               os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
