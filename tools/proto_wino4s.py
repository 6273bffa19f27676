"""Prototype of VERDICT r4 item 1: the fp32 F(4x4) forward / dgrad convolution on the bf16 matrix pipe with a three-term
(six-product) split -- tools/proto/conv3x3_wino4s.hip, built here into tools/proto/libwino4s.so (NOT part of libaide_hip.so:
the prototype did not reach the 1.2x-alone bar, profiles/r05_split_proto.md) -- against the shipped fp32 F(4x4) kernel:
error of both vs float64 of the same fp32 inputs, and time alone.
Usage: python tools/proto_wino4s.py [--quick] [--build-only]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes
import subprocess
from aide_amd import ops
from aide_amd._lib import lib, check
from aide_amd.ops import ptr, planes, stream_ptr

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'proto', 'conv3x3_wino4s.hip')
SO = os.path.join(HERE, 'proto', 'libwino4s.so')


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        lib.load()
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast',
                               '-shared', SRC, '-o', SO, '-L' + os.path.join(os.path.dirname(HERE), 'aide_amd'), '-laide_hip',
                               '-Wl,-rpath,' + os.path.join(os.path.dirname(HERE), 'aide_amd')])
    lib.load()                                   # aide_ktimer_slot lives in libaide_hip.so
    dll = ctypes.CDLL(SO)
    fn = dll.aide_conv3x3_wino4s
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64] + \
        [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return fn


def wino4s_pack_torch(w, dgrad=False):
    """Split-bf16 F(4x4) filter pack in plain torch (prototype / test reference of the device pack): w [Co,Ci,3,3] ->
    u [Ci/8][36][4 planes uh, um, ul, uh][Co][8] bf16 as int16 (dgrad: roles of Co / Ci swapped, taps reversed)."""
    if dgrad:
        w = w.flip(2, 3).transpose(0, 1)
    co, ci = w.shape[0], w.shape[1]
    G = torch.tensor([[0.25, 0, 0], [-1 / 6., -1 / 6., -1 / 6.], [-1 / 6., 1 / 6., -1 / 6.], [1 / 24., 1 / 12., 1 / 6.],
                      [1 / 24., -1 / 12., 1 / 6.], [0, 0, 1]], dtype=torch.float64, device=w.device)
    u = torch.einsum('ar,ocrs,bs->aboc', G, w.double(), G).float()            # [6][6][Co][Ci]
    slot = [4, 0, 1, 2, 3, 5]
    order = [0] * 36
    for r in range(6):
        for c in range(6):
            order[18 * (r // 3) + 6 * (r % 3) + slot[c]] = 6 * r + c
    u = u.reshape(36, co, ci)[order]                                             # [vp][Co][Ci]
    uh = u.bfloat16()
    r1 = u - uh.float()
    um = r1.bfloat16()
    ul = (r1 - um.float()).bfloat16()
    t = torch.stack([uh, um, ul, uh], 1)                                         # [vp][4][Co][Ci]
    t = t.reshape(36, 4, co, ci // 8, 2, 4).permute(3, 0, 1, 2, 5, 4).contiguous()   # [s][vp][4][Co][d][e]
    return t.view(torch.int16).reshape(ci // 8, 36, 4, co, 8)



def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run_s(x, u, bias, y, splitk, ws):
    xp, xbs = planes(x)
    yp, ybs = planes(y)
    n, cin, h, w = x.shape
    cout = y.shape[1]
    check(W4S(xp, xbs, ptr(u), ptr(bias), yp, ybs, n, cin, h, w, cout, 0, splitk, ptr(ws), None, stream_ptr()), 'conv3x3_wino4s')


def main():
    global W4S
    W4S = build()
    if '--build-only' in sys.argv:
        return
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    shapes = [(4, 128, 64, 256), (4, 512, 256, 64), (4, 64, 64, 256), (4, 256, 128, 128), (4, 1024, 512, 32)]
    if '--quick' in sys.argv:
        shapes = shapes[:2]
    for n, ci, co, h in shapes:
        x = torch.randn(n, ci, h, h, device=dev)
        x = torch.relu(x) + 0.1 * x            # activation-like (mostly positive) inputs
        w = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
        b = torch.randn(co, device=dev)
        y4 = torch.empty(n, co, h, h, device=dev)
        ys = torch.empty(n, co, h, h, device=dev)
        uf4, _ = ops.wino4_pack(w, need_dgrad=False)
        us = wino4s_pack_torch(w)
        gf = 2.0 * n * h * h * co * ci * 9 / 1e9
        sk4 = lib.aide_conv3x3_wino4_splitk(n, ci, h, h, co)
        ws = torch.empty(max(1, sk4) * n * co * h * h, device=dev)
        ops.conv3x3_wino4(x, uf4, b, y4, splitk=1)
        run_s(x, us, b, ys, 1, None)
        torch.cuda.synchronize()
        # float64 reference of image 0 on the CPU
        ref = torch.nn.functional.conv2d(x[:1].double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
        sc = ref.abs().max().item()
        e4 = (y4[:1].double().cpu() - ref).abs().max().item() / sc
        es = (ys[:1].double().cpu() - ref).abs().max().item() / sc
        r4 = ((y4[:1].double().cpu() - ref) ** 2).mean().sqrt().item() / sc
        rs = ((ys[:1].double().cpu() - ref) ** 2).mean().sqrt().item() / sc
        d = (ys - y4).abs().max().item() / sc
        line = '%4d->%4d @%3d N%d %7.2f GF | err/max|y| vs f64: fp32 F4 %.2e (rms %.2e)  split %.2e (rms %.2e)  split-fp32 %.2e |' % (
            ci, co, h, n, gf, e4, r4, es, rs, d)
        for sk in sorted(set([1, sk4])):
            if (ci // 16) % sk:
                continue
            acc = 2 if sk > 1 else 0            # slabs only: the reduce is the same kernel for both
            xp, xbs = planes(x)
            yp, ybs = planes(y4)
            t4 = timeit(lambda: check(lib.aide_conv3x3_wino4(xp, xbs, ptr(uf4), ptr(b), yp, ybs, n, ci, h, h, co, acc, sk,
                                                             ptr(ws), None, None, 0, None, 0, stream_ptr()), 'w4'))
            ts = timeit(lambda: run_s(x, us, b, ys, sk, ws))
            line += ' s%d: fp32 %.3f ms %.1f TF, split %.3f ms %.1f TF = %.2fx |' % (sk, t4, gf / t4, ts, gf / ts, t4 / ts)
        print(line, flush=True)


if __name__ == '__main__':
    main()
