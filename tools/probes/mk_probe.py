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

# ---- conv3x3_wgrad4_kernel: which LDS access class carries the bank conflicts
G4_PUT = """                    if (3 * (st - 31) + q < 14) put_d(3 * (st - 31) + q, rawc);
"""
G4_VSTORE = "            if (st >= 15 && st < 24) v_store(st - 15, 1 - kcur);\n"
G4_ZSTORE = """                for (int q = 0; q < 2; ++q) z_store(2 * (st - 21) + q, 1 - kcur);
"""
G4_ZNONE = "                for (int q = 0; q < 2; ++q) { }\n"
G4_VREAD = "            if (st >= 1 && st < 7) v_read(st - 1, kcur ? xr0 : xr1);\n"
G4_FRAG = "            if (w == 0 && gq + 2 < 9) frag(gq + 2, (gq + 2) % 3);\n"

# name -> (source file, [(old, new), ...])
BN_LIM = 'constexpr long BN_ONEPASS_MAX_VALUES = 34L << 20, BN_ONEPASS_MAX_VALUES_NARROW = 9L << 20;'
PROBES = {
    # round 6: layers of >= 256 (co, ci) tiles as consecutive half-chip launches instead of one whole-chip launch (measured: loses)
    'g4_split': ('conv3x3_wgrad4.hip', [("    const long per_launch = nb;\n", "    const long target = target_wgs > 0 ? target_wgs : 128;\n    const long per_launch = (target < 256 && nb >= 2 * target) ? target : nb;\n")]),
    # one-pass BatchNorm: values per thread (16 shipped) and the launch size below which a thread takes fewer
    'bn_v32': ('bn.hip', [("int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);           // 32 values per thread ...", "int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);")]),
    'bn_v64': ('bn.hip', [("int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);           // 32 values per thread ...", "int q = 64 / p.V, s = (units + 256 * q - 1) / (256 * q);")]),
    'bn_v32w512': ('bn.hip', [("int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);           // 32 values per thread ...", "int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);"), ("constexpr int BN_COOP_MIN_WGS = 512;", "constexpr int BN_COOP_MIN_WGS = 512;")]),
    'bn_v64w512': ('bn.hip', [("int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);           // 32 values per thread ...", "int q = 64 / p.V, s = (units + 256 * q - 1) / (256 * q);"), ("constexpr int BN_COOP_MIN_WGS = 512;", "constexpr int BN_COOP_MIN_WGS = 512;")]),
    'bn_v8': ('bn.hip', [("int q = 32 / p.V, s = (units + 256 * q - 1) / (256 * q);           // 32 values per thread ...", "int q = 8 / p.V, s = (units + 256 * q - 1) / (256 * q);")]),
    'bn_wg2048': ('bn.hip', [("constexpr int BN_COOP_MIN_WGS = 512;", "constexpr int BN_COOP_MIN_WGS = 2048;")]),
    'bn_wg512': ('bn.hip', [("constexpr int BN_COOP_MIN_WGS = 512;", "constexpr int BN_COOP_MIN_WGS = 512;")]),
    'bn_sp4096': ('bn.hip', [("    int s = (int)((2048 + C - 1) / C);", "    int s = (int)((4096 + C - 1) / C);")]),
    'bn_sp1024': ('bn.hip', [("    int s = (int)((2048 + C - 1) / C);", "    int s = (int)((1024 + C - 1) / C);")]),
    'bn_lead0': ('bn.hip', [("constexpr bool BN_LEADER_LAST = true;", "constexpr bool BN_LEADER_LAST = false;")]),
    'bn_sleep1': ('bn.hip', [("        __builtin_amdgcn_s_sleep(4);\n", "        __builtin_amdgcn_s_sleep(1);\n")]),
    'bn_sleep16': ('bn.hip', [("        __builtin_amdgcn_s_sleep(4);\n", "        __builtin_amdgcn_s_sleep(16);\n")]),
    # one-pass BatchNorm only below a tensor size (round 6): 0 = the two-pass kernels everywhere
    'bn_2pass': ('bn.hip', [(BN_LIM, 'constexpr long BN_ONEPASS_MAX_VALUES = 0, BN_ONEPASS_MAX_VALUES_NARROW = 0;')]),
    'bn_lim8m': ('bn.hip', [(BN_LIM, 'constexpr long BN_ONEPASS_MAX_VALUES = 9L << 20, BN_ONEPASS_MAX_VALUES_NARROW = 9L << 20;')]),
    'bn_lim17m': ('bn.hip', [(BN_LIM, 'constexpr long BN_ONEPASS_MAX_VALUES = 17L << 20, BN_ONEPASS_MAX_VALUES_NARROW = 17L << 20;')]),
    'bn_lim34m': ('bn.hip', [(BN_LIM, 'constexpr long BN_ONEPASS_MAX_VALUES = 34L << 20, BN_ONEPASS_MAX_VALUES_NARROW = 34L << 20;')]),
    # hand-over events with a device-scope release instead of the default (round 5: the 5-7 us hole behind every kernel that carries a
    # completion event in the backward pass -- is it the system-scope fence of the event?)
    'ev_device': ('ktimer.hip', [("hipEventCreateWithFlags(&e, hipEventDisableTiming);", "hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice);")]),
    'ev_nofence': ('ktimer.hip', [("hipEventCreateWithFlags(&e, hipEventDisableTiming);", "hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);")]),
    'g4_noput': ('conv3x3_wgrad4.hip', [(G4_PUT, '                    { }\n')]),                 # no raw-tile stores (14 ds_write_b32)
    'g4_novstore': ('conv3x3_wgrad4.hip', [(G4_VSTORE, '')]),                              # no V stores (9 ds_write_b64)
    'g4_nozstore': ('conv3x3_wgrad4.hip', [(G4_ZSTORE, G4_ZNONE)]),          # no ZT stores (12 b64 + 12 b32)
    'g4_novread': ('conv3x3_wgrad4.hip', [(G4_VREAD, '')]),                                # no patch reads (6 x b128 + b64)
    'g4_nofrag': ('conv3x3_wgrad4.hip', [(G4_FRAG, '')]),                                  # fragment reads of the first two pairs only
    # V stores as single ds_write_b64 (inline asm: hipcc cannot pair them into ds_write2st64_b64) -- are the conflicts the
    # counter attributes to the V stores a property of the paired form (both halves of a pair hit the same banks)?
    'g4_vunmerged': ('conv3x3_wgrad4.hip', [
        ("    auto v_store = [&](int m, int set) { lds2[(set ? vb1 : vb0) + m * 128] = tq[m]; };\n",
         "    auto v_store = [&](int m, int set) { asm volatile(\"ds_write_b64 %0, %1 offset:%2\" :: \"v\"((set ? vb1 : vb0) * 8), \"v\"(tq[m]), \"i\"(m * 1024) : \"memory\"); };\n")]),
    # default workgroup count of the F(4x4) weight gradient (it shares the chip with the dependent chain)
    'g4_t112': ('conv3x3_wgrad4.hip', [("target_wgs > 0 ? target_wgs : 128;", "target_wgs > 0 ? target_wgs : 112;")]),
    'g4_t144': ('conv3x3_wgrad4.hip', [("target_wgs > 0 ? target_wgs : 128;", "target_wgs > 0 ? target_wgs : 144;")]),
    'g4_t160': ('conv3x3_wgrad4.hip', [("target_wgs > 0 ? target_wgs : 128;", "target_wgs > 0 ? target_wgs : 160;")]),
    # workgroup target of the bf16 weight gradient (its 128-co tile fills a CU: what it leaves is what the main stream gets)
    'wb_t128': ('conv3x3_bf16.hip', [("const long target = tco == 32 ? 512 : 192;", "const long target = tco == 32 ? 512 : 128;")]),
    'wb_t144': ('conv3x3_bf16.hip', [("const long target = tco == 32 ? 512 : 192;", "const long target = tco == 32 ? 512 : 144;")]),
    'wb_t160': ('conv3x3_bf16.hip', [("const long target = tco == 32 ? 512 : 192;", "const long target = tco == 32 ? 512 : 160;")]),
    'wb_t176': ('conv3x3_bf16.hip', [("const long target = tco == 32 ? 512 : 192;", "const long target = tco == 32 ? 512 : 176;")]),
    'wb_t208': ('conv3x3_bf16.hip', [("const long target = tco == 32 ? 512 : 192;", "const long target = tco == 32 ? 512 : 208;")]),
    'g4_nostores': ('conv3x3_wgrad4.hip', [(G4_PUT, '                    { }\n'), (G4_VSTORE, ''), (G4_ZSTORE, G4_ZNONE)]),
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
