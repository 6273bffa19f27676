"""Micro-benchmark of the conv kernels on the FuseUNet C2 layer shapes (N=4, 256x256).
Usage: python tools/bench_conv.py [fwd|dgrad|wgrad|all] [--variants]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aide_amd import ops
from aide_amd._lib import lib

SHAPES = [  # Ci, Co, H, count
    (3, 32, 256, 2), (32, 32, 256, 2), (64, 64, 128, 3), (32, 64, 128, 1), (128, 128, 64, 3),
    (64, 128, 64, 1), (256, 256, 32, 3), (128, 256, 32, 1), (512, 512, 16, 3), (256, 512, 16, 1),
    (1024, 512, 32, 2), (512, 512, 32, 1), (512, 256, 64, 2), (256, 256, 64, 1), (256, 128, 128, 2),
    (128, 128, 128, 1), (128, 64, 256, 2), (64, 64, 256, 1)]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
    sweep = '--variants' in sys.argv
    dev = torch.device('cuda:0')
    N = 4
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    totf = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    for ci, co, h, cnt in SHAPES:
        x = torch.randn(N, ci, h, h, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        dy = torch.randn(N, co, h, h, device=dev)
        y = torch.empty(N, co, h, h, device=dev)
        dx = torch.empty(N, ci, h, h, device=dev)
        dw = torch.empty(co, ci, 3, 3, device=dev)
        wf, wd = ops.pack_weights(w, need_dgrad=ci % 32 == 0)
        gf = 2.0 * N * h * h * co * ci * 9 / 1e9
        line = '%4d->%4d @%3d x%d  %7.2f GF |' % (ci, co, h, cnt, gf)
        if mode in ('fwd', 'all'):
            plan = lib.aide_conv3x3_plan(N, ci, h, h, co)
            ws = torch.empty(max(1, lib.aide_conv3x3_ws_bytes(N, h, h, co, plan >> 8) // 4), device=dev)
            t = timeit(lambda: ops.conv3x3_igemm(x, wf, b, y, plan=plan, ws=ws))
            line += ' fwd v%d/s%d %7.3f ms %6.1f TF |' % (plan & 255, plan >> 8, t, gf / t)
            tot['fwd'] += t * cnt; totf['fwd'] += gf * cnt
            if sweep:
                for v in range(6):
                    if (v in (2, 5) and co % 128) or (v != 0 and co % 64) or (ci < 8 and v > 1):
                        continue
                    for sk in (1, 2, 4, 8):
                        if sk > 1 and (ci // 8) < sk * 4:
                            continue
                        ws2 = torch.empty(max(1, lib.aide_conv3x3_ws_bytes(N, h, h, co, sk) // 4), device=dev)
                        t2 = timeit(lambda: ops.conv3x3_igemm(x, wf, b, y, plan=v | (sk << 8), ws=ws2), 5)
                        line += '\n      v%d s%d %7.3f ms %6.1f TF' % (v, sk, t2, gf / t2)
        if mode in ('dgrad', 'all') and wd is not None:
            plan = lib.aide_conv3x3_plan(N, co, h, h, ci)
            ws = torch.empty(max(1, lib.aide_conv3x3_ws_bytes(N, h, h, ci, plan >> 8) // 4), device=dev)
            t = timeit(lambda: ops.conv3x3_igemm(dy, wd, None, dx, plan=plan, ws=ws))
            line += ' dgrad v%d/s%d %7.3f ms %6.1f TF |' % (plan & 255, plan >> 8, t, gf / t)
            tot['dgrad'] += t * cnt; totf['dgrad'] += gf * cnt
        if mode in ('wino', 'all') and ops.wino_supported(ci, h, h, co):
            uf, ud = ops.wino_pack(w, need_dgrad=ci % 64 == 0)
            wsw = torch.empty(1 << 26, device=dev)
            sk = lib.aide_conv3x3_wino_splitk(N, ci, h, h, co)
            t = timeit(lambda: ops.conv3x3_wino(x, uf, b, y, splitk=sk, ws=wsw))
            line += ' WINO fwd s%d %7.3f ms %6.1f TF |' % (sk, t, gf / t)
            tot.setdefault('wino', 0.0); totf.setdefault('wino', 0.0)
            tot['wino'] += t * cnt; totf['wino'] += gf * cnt
            if ud is not None:
                sk = lib.aide_conv3x3_wino_splitk(N, co, h, h, ci)
                t = timeit(lambda: ops.conv3x3_wino(dy, ud, None, dx, splitk=sk, ws=wsw))
                line += ' WINO dgrad s%d %7.3f ms %6.1f TF |' % (sk, t, gf / t)
        if mode in ('wino4', 'wino', 'all') and ops.wino4_supported(ci, h, h, co):
            uf4, ud4 = ops.wino4_pack(w, need_dgrad=ci % 64 == 0)
            wsw = torch.empty(1 << 26, device=dev)
            for sk in sorted(set([lib.aide_conv3x3_wino4_splitk(N, ci, h, h, co)] + ([1, 2, 4, 8] if sweep else []))):
                if (ci // 8) % sk:
                    continue
                t = timeit(lambda: ops.conv3x3_wino4(x, uf4, b, y, splitk=sk, ws=wsw))
                line += ' F4 fwd s%d %7.3f ms %6.1f TF |' % (sk, t, gf / t)
            tot.setdefault('wino4', 0.0); totf.setdefault('wino4', 0.0)
            tot['wino4'] += t * cnt; totf['wino4'] += gf * cnt
            if ud4 is not None:
                sk = lib.aide_conv3x3_wino4_splitk(N, co, h, h, ci)
                t = timeit(lambda: ops.conv3x3_wino4(dy, ud4, None, dx, splitk=sk, ws=wsw))
                line += ' F4 dgrad s%d %7.3f ms %6.1f TF |' % (sk, t, gf / t)
        if mode in ('wgrad', 'all'):
            ws = torch.empty(lib.aide_conv3x3_wgrad_ws_bytes(N, co, ci, h, h) // 4, device=dev)
            t = timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, ws=ws))
            line += ' wgrad s%d %7.3f ms %6.1f TF' % (lib.aide_conv3x3_wgrad_splits(N, co, ci, h, h), t, gf / t)
            if ops.wgrad_wino_supported(co, ci, h, h):
                ws2 = torch.empty(lib.aide_conv3x3_wgrad_wino_ws_bytes(N, co, ci, h, h) // 4, device=dev)
                t2 = timeit(lambda: ops.conv3x3_wgrad_wino(dy, x, dw, ws=ws2))
                line += ' | WINO wgrad s%d %7.3f ms %6.1f TF' % (lib.aide_conv3x3_wgrad_wino_splits(N, co, ci, h, h), t2, gf / t2)
                tot.setdefault('wino_wgrad', 0.0); totf.setdefault('wino_wgrad', 0.0)
                tot['wino_wgrad'] += t2 * cnt; totf['wino_wgrad'] += gf * cnt
            if ops.wgrad_wino4_supported(co, ci, h, h):
                ws4 = torch.empty(lib.aide_conv3x3_wgrad_wino4_ws_bytes(N, co, ci, h, h) // 4, device=dev)
                t4 = timeit(lambda: ops.conv3x3_wgrad_wino4(dy, x, dw, ws=ws4))
                line += ' | F4 wgrad s%d %7.3f ms %6.1f TF' % (lib.aide_conv3x3_wgrad_wino4_splits(N, co, ci, h, h), t4, gf / t4)
                tot.setdefault('wgrad4', 0.0); totf.setdefault('wgrad4', 0.0)
                tot['wgrad4'] += t4 * cnt; totf['wgrad4'] += gf * cnt
            tot['wgrad'] += t * cnt; totf['wgrad'] += gf * cnt
        print(line, flush=True)
    for k in tot:
        if tot[k] > 0:
            print('%s total: %.3f ms, %.1f GF, %.1f TF/s' % (k, tot[k], totf[k], totf[k] / tot[k]))


if __name__ == '__main__':
    main()
