"""Build libktup_hip.so (the C-ABI library) in-tree with hipcc for gfx950.

    python joint-kg-recommender_amd/build_hip.py [--force] [--save-temps]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the tree.
It links against libamdhip64.so.7; inside a Python process that imported torch first, the dynamic
loader resolves that SONAME to torch's bundled HIP runtime, so streams and device pointers are shared.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libktup_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function']
# SLP packs scalar-operand FMAs into v_pk_fma_f32, which needs SGPR operands in aligned pairs (one s_mov per FMA): off
# for the kernels that feed FMAs from SGPRs.
PER_FILE_FLAGS = {'ktup_score_pref.hip': ['-fno-slp-vectorize'], 'ktup_eval.hip': ['-fno-slp-vectorize']}


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
        [os.path.join(os.path.dirname(HERE), 'include', 'ktup_hip.h')]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, extra):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
    cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + extra + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    return src, obj, r.returncode, r.stdout + r.stderr


def build(force=False, save_temps=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs, hdrs = _sources(), _headers()
    extra = ['-save-temps'] if save_temps else []
    todo = [s for s in srcs if force or _stale(os.path.join(OBJ, os.path.basename(s)[:-4] + '.o'), [s] + hdrs)]
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + '.o') for s in srcs]
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            for src, obj, rc, out in ex.map(lambda s: _compile(s, extra), todo):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError('hipcc failed on %s\n%s' % (src, out))
                if verbose:
                    print('compiled', os.path.basename(src))
    if todo or force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed\n' + r.stdout + r.stderr)
        if verbose:
            print('linked', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, save_temps='--save-temps' in sys.argv)
