"""Build libmasr_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

Every source is compiled to its own object (re-compiled only when it, a shared header or the public header is newer), the
objects are compiled in parallel, and the link step runs when any object changed."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, 'csrc')
LIB_DIR = os.path.join(ROOT, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libmasr_hip.so')
SOURCES = ['gemm_f32.hip', 'ffn_reduce.hip', 'ffn_pc.hip', 'sqz_layer.hip', 'rowgemm.hip', 'rowgemm_small.hip', 'elementwise.hip',
           'attention.hip', 'lstm.hip', 'beam_gpu.hip', 'lm_scorer.cpp', 'fbank.hip', 'silero.hip', 'engine.hip', 'pool.hip',
           'beam_search.cpp', 'stage.cpp', 'resample.cpp']
# MASR_BUILD_EXPERIMENTS=1: the measured-and-rejected kernels of earlier rounds (A/B material behind masr_debug_set keys 20 / 24 /
# 30 / 34 / 35) are compiled in as well; the default library holds the product kernels only
EXPERIMENTS = os.environ.get('MASR_BUILD_EXPERIMENTS') == '1'
EXPERIMENT_SOURCES = ['gemm_bf16x3.hip', 'ffn_x3.hip', 'ffn_coop.hip', 'ffn_dual.hip']
if EXPERIMENTS:
    SOURCES = SOURCES + EXPERIMENT_SOURCES
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'lm_scorer.h'),
           os.path.join(os.path.dirname(ROOT), 'include', 'masr_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value'] + (['-DMASR_EXPERIMENTS=1'] if EXPERIMENTS else [])
STAMP = os.path.join(LIB_DIR, '.build_flavour')


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _obj(src):
    return os.path.join(LIB_DIR, os.path.splitext(src)[0] + '.o')


def _stale(src):
    obj = _obj(src)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src)] + [h for h in HEADERS if os.path.exists(h)]
    return any(os.path.getmtime(d) > t for d in deps)


def _flavour():
    return 'experiments' if EXPERIMENTS else 'product'


def _flavour_changed():
    try:
        return open(STAMP).read().strip() != _flavour()
    except OSError:
        return os.path.exists(LIB_PATH)          # a library of unknown flavour


def needs_build():
    if not os.path.exists(LIB_PATH) or _flavour_changed():
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(_stale(s) or os.path.getmtime(_obj(s)) > t for s in _sources())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> masr_amd/lib/libmasr_hip.so"""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    force = force or _flavour_changed()             # objects of the other flavour were compiled with other macros
    todo = [s for s in _sources() if force or _stale(s)]

    def compile_one(src):
        cmd = [_hipcc()] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', _obj(src)]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', LIB_PATH] + [_obj(s) for s in _sources()]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, 'w') as f:
        f.write(_flavour() + '\n')
    return LIB_PATH


def has_experiments():
    """flavour of the library on disk (tests of the experimental kernels skip on the product build)"""
    try:
        return open(STAMP).read().strip() == 'experiments'
    except OSError:
        return False


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
