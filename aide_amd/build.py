"""Build libaide_hip.so (gfx950) in-tree with hipcc. No GPU is needed to compile.

    python -m aide_amd.build [--force]

The .so stays inside aide_amd/ (git-ignored, but it travels with the tree to the GPU box).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libaide_hip.so')
SOURCES = ['conv3x3.hip', 'conv3x3_winograd.hip', 'conv3x3_wino4.hip', 'conv3x3_wgrad.hip', 'conv3x3_wgrad4.hip', 'conv3x3_wgrad_stem.hip', 'conv3x3_bf16.hip', 'bn.hip', 'spatial.hip', 'loss.hip', 'loss_mc.hip', 'coteach_ext.hip', 'head_adam.hip',
           'convt.hip', 'attention.hip', 'ktimer.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _digest(paths):
    h = hashlib.sha1()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    common = os.path.join(CSRC, 'common.h')

    def compile_one(src):
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.replace('.hip', '.o'))
        stamp = op + '.sha1'
        dg = _digest([sp, common])
        if not force and os.path.exists(op) and os.path.exists(stamp) and open(stamp).read() == dg:
            return op, False
        cmd = [hipcc] + FLAGS + ['-c', sp, '-o', op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr))
        with open(stamp, 'w') as f:
            f.write(dg)
        return op, True

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or not os.path.exists(LIB):
        r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr)
        if verbose:
            print('built', LIB)
    elif verbose:
        print('up to date', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
