"""Layer sweep of the bf16-MFMA conv kernels at the FuseUNet 512x512 bs=8 (BASELINE config 5) layer shapes,
next to the fp32 path the engine would otherwise use.  python tools/bench_bf16.py [c5|c2] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd import ops          # noqa: E402
from aide_amd._lib import lib     # noqa: E402

# (Cin, Cout, level) of the distinct FuseUNet conv shapes (fuseunet.py:12-39)
LAYERS = [(3, 32, 0), (32, 32, 0), (64, 64, 1), (32, 64, 1), (128, 128, 2), (64, 128, 2), (256, 256, 3),
          (128, 256, 3), (512, 512, 4), (256, 512, 4), (1024, 512, 3), (512, 256, 2), (256, 128, 1),
          (128, 64, 0), (64, 64, 0), (512, 512, 3), (256, 256, 2), (128, 128, 1)]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n, size = (8, 512) if cfg == 'c5' else (4, 256)
    dev = torch.device('cuda:0')
    tot = {'fwd': [0.0, 0.0], 'dgrad': [0.0, 0.0], 'wgrad': [0.0, 0.0]}
    print('%-22s %10s %10s %10s   (ms, TFLOP/s algorithmic)' % ('layer', 'fwd', 'dgrad', 'wgrad'))
    only = os.environ.get('AIDE_ONLY')          # e.g. AIDE_ONLY=128,64,0 -> one layer (PMC runs)
    layers = [tuple(int(v) for v in only.split(','))] if only else LAYERS
    for ci, co, lv in layers:
        h = w = size >> lv
        if not ops.bf16_supported(ci, h, w, co):
            print('%4d->%4d @%3d  unsupported' % (ci, co, h))
            continue
        store = os.environ.get('AIDE_SWEEP_STORE', '1') != '0' and ci % 16 == 0     # bf16-stored tensors (engine default)
        sdt = torch.bfloat16 if store else torch.float32
        x = torch.randn(n, ci, h, w, device=dev).to(sdt)
        dy = torch.randn(n, co, h, w, device=dev).to(sdt)
        wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        y = torch.empty(n, co, h, w, device=dev, dtype=torch.bfloat16 if os.environ.get('AIDE_SWEEP_STORE', '1') != '0' else torch.float32)
        dx = torch.empty(n, ci, h, w, device=dev, dtype=sdt)
        dw = torch.empty_like(wt)
        uf, ud = ops.bf16_pack(wt, need_dgrad=ci % 32 == 0)
        flops = 2.0 * n * h * w * ci * co * 9
        skf = lib.aide_conv3x3_bf16_splitk(n, ci, h, w, co)
        wsf = torch.empty(max(lib.aide_conv3x3_ws_bytes(n, h, w, max(co, ci), 16) // 4, 1), device=dev)
        row = '%4d->%4d @%3d sk%-2d' % (ci, co, h, skf)
        t = timeit(lambda: ops.conv3x3_bf16(x, uf, b, y, splitk=skf, ws=wsf), reps)
        tot['fwd'][0] += t; tot['fwd'][1] += flops
        row += '  %6.3f %5.0f' % (t, flops / t * 1e-9)
        if ud is not None:
            skd = lib.aide_conv3x3_bf16_splitk(n, co, h, w, ci)
            t = timeit(lambda: ops.conv3x3_bf16(dy, ud, None, dx, splitk=skd, ws=wsf), reps)
            tot['dgrad'][0] += t; tot['dgrad'][1] += flops
            row += '  %6.3f %5.0f' % (t, flops / t * 1e-9)
        else:
            row += '  %12s' % '-'
        if ops.wgrad_bf16_supported(co, ci, h, w):
            ws = torch.empty(lib.aide_conv3x3_wgrad_bf16_ws_bytes(n, co, ci, h, w, 0) // 4, device=dev)
            t = timeit(lambda: ops.conv3x3_wgrad_bf16(dy, x, dw, ws=ws), reps)
            tot['wgrad'][0] += t; tot['wgrad'][1] += flops
            row += '  %6.3f %5.0f (splits %d)' % (t, flops / t * 1e-9, lib.aide_conv3x3_wgrad_bf16_splits(n, co, ci, h, w, 0))
        print(row, flush=True)
        del x, dy, y, dx, wsf
    for k, (t, f) in tot.items():
        if t:
            print('%-6s total %.3f ms over distinct shapes, %.0f TFLOP/s' % (k, t, f / t * 1e-9))


if __name__ == '__main__':
    main()
