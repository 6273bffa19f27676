#!/usr/bin/env python
"""Ablation builds WITHOUT probe macros in the shipped sources: a named list of text edits is applied to a COPY of one
csrc/*.hip, the copy is compiled and linked against the objects of the current build -> abtest/lib_<name>.so (selected at
run time with AIDE_HIP_LIB).  Timing probes only: most of them compute wrong results.

    python tools/probes/mk_probe.py <name> [<name> ...]      (names: the keys of PROBES)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'aide_amd', 'csrc')
OBJ = os.path.join(ROOT, 'aide_amd', 'build')
OUT = os.path.join(ROOT, 'abtest')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast']

WG_MFMA = """                acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xf[kh], acc[kh * 3 + 0], 0, 0, 0);
                acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xf[kh], acc[kh * 3 + 1], 0, 0, 0);
                acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xf[kh], acc[kh * 3 + 2], 0, 0, 0);
"""
WG_NOMFMA = """                acc[kh * 3 + 0][0] += __builtin_bit_cast(f32x4, a0)[0] + __builtin_bit_cast(f32x4, xf[kh])[0];
                acc[kh * 3 + 1][1] += __builtin_bit_cast(f32x4, a1)[1] + __builtin_bit_cast(f32x4, xf[kh])[1];
                acc[kh * 3 + 2][2] += __builtin_bit_cast(f32x4, a2)[2] + __builtin_bit_cast(f32x4, xf[kh])[2];
"""
WG_PUT = "                if (op < NOPS) put(op, nxt);\n"
WG_FETCH = "                if (op < NOPS) fetch(op);\n"
WG_BARRIER = "            __builtin_amdgcn_sched_barrier(0);\n        }\n        __syncthreads();\n        cur ^= 1;\n"

# name -> (source file, [(old, new), ...])
PROBES = {
    'wg_nomfma': ('conv3x3_bf16.hip', [(WG_MFMA, WG_NOMFMA)]),
    'wg_nofetch': ('conv3x3_bf16.hip', [(WG_FETCH, '')]),
    'wg_noput': ('conv3x3_bf16.hip', [(WG_PUT, '')]),
    'wg_nostage': ('conv3x3_bf16.hip', [(WG_FETCH, ''), (WG_PUT, '')]),
    'wg_nobarrier': ('conv3x3_bf16.hip', [(WG_BARRIER, "            __builtin_amdgcn_sched_barrier(0);\n        }\n        cur ^= 1;\n")]),
    'wg_mfmaonly': ('conv3x3_bf16.hip', [(WG_FETCH, ''), (WG_PUT, ''),
                                         (WG_BARRIER, "            __builtin_amdgcn_sched_barrier(0);\n        }\n        cur ^= 1;\n")]),
}


def build(name):
    src, edits = PROBES[name]
    text = open(os.path.join(CSRC, src)).read()
    for old, new in edits:
        if text.count(old) != 1:
            raise SystemExit('%s: edit anchor found %d times in %s:\n%s' % (name, text.count(old), src, old))
        text = text.replace(old, new)
    os.makedirs(OUT, exist_ok=True)
    cp = os.path.join(OUT, '%s__%s' % (name, src))
    with open(cp, 'w') as f:
        f.write(text)
    obj = cp.replace('.hip', '.o')
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + ['-I', CSRC, '-c', cp, '-o', obj])
    others = [os.path.join(OBJ, o) for o in sorted(os.listdir(OBJ)) if o.endswith('.o') and o != src.replace('.hip', '.o')]
    lib = os.path.join(OUT, 'lib_%s.so' % name)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, obj] + others)
    print('built', lib)


if __name__ == '__main__':
    for n in (sys.argv[1:] or sorted(PROBES)):
        build(n)
