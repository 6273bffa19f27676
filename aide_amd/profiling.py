"""Per-kernel timing with HIP events recorded on the launch stream (used by bench.py's roofline)."""
import collections

import torch


class KernelTimer(object):
    """begin(tag, flops, executed) / end() bracket ONE kernel-family launch with a HIP event pair on the
    current stream. `flops` is the algorithmic (direct-convolution) count, `executed` what the MFMA pipe
    really multiplies (16/36 of it for a Winograd F(2x2,3x3) launch). Events are resolved after a device
    synchronise by summary()."""

    def __init__(self, reserve=0):
        self.records = []
        self._open = None
        # events are created up front: creating them inside the timed region costs far more than recording them
        # (hundreds of hipEventCreate calls stalled the launch queue by ~14 ms per instrumented step)
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reserve)]
        for e in self._pool:             # first record of an event allocates its completion signal: do it out here too
            e.record()
        if self._pool:
            torch.cuda.synchronize()

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def begin(self, tag, flops, executed=None):
        e0 = self._event()
        e0.record()
        self._open = (tag, flops, flops if executed is None else executed, e0)

    def end(self):
        tag, flops, executed, e0 = self._open
        e1 = self._event()
        e1.record()
        self.records.append((tag, flops, executed, e0, e1))
        self._open = None

    def summary(self):
        torch.cuda.synchronize()
        agg = collections.OrderedDict()
        for tag, flops, executed, e0, e1 in self.records:
            a = agg.setdefault(tag, dict(launches=0, ms=0.0, flops=0.0, executed=0.0))
            a['launches'] += 1
            a['ms'] += e0.elapsed_time(e1)
            a['flops'] += flops
            a['executed'] += executed
        for a in agg.values():
            a['avg_ms'] = a['ms'] / max(a['launches'], 1)
            a['tflops'] = a['flops'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] > 0 else 0.0
            a['executed_tflops'] = a['executed'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] > 0 else 0.0
        return agg


# family ids of the C-side kernel timer (include/aide_hip.h "kernel timer") -> (kernel name, executed / algorithmic multiplies)
KT_FAMILIES = {
    0: ('conv3x3_mfma_kernel', 1.0), 1: ('conv3x3_wino_kernel', 16.0 / 36.0), 2: ('conv3x3_wino4_kernel', 36.0 / 144.0),
    3: ('conv3x3_wgrad_kernel', 1.0), 4: ('conv3x3_wgrad_wino_kernel', 16.0 / 36.0),
    5: ('conv3x3_wgrad4_kernel', 36.0 / 144.0), 6: ('wgrad_stem_kernel', 1.0), 7: ('conv3x3_bf16_kernel', 1.0),
    8: ('conv3x3_wgrad_bf16_kernel', 1.0),
}


# the streaming families (work = algorithmic bytes per launch) and the rest of the step (work 0)
KT_STREAMING = {10: 'batchnorm_fwd', 11: 'batchnorm_bwd', 12: 'maxpool', 13: 'upsample'}
KT_REST = {9: 'convT', 14: 'splitk_slab_reduce', 15: 'head_loss_adam_pack'}
KT_ALL = dict([(k, v[0]) for k, v in KT_FAMILIES.items()] + list(KT_STREAMING.items()) + list(KT_REST.items()))


class DispatchTimer(object):
    """Per-dispatch timing of the MFMA convolution kernels through the C ABI's kernel timer: while armed, every launch of
    a family's main kernel carries a hipExtLaunchKernelGGL start / stop event pair = the dispatch's own begin / end
    timestamps (the duration rocprofv3 --kernel-trace reports), on whatever stream it runs -- the timed steps keep
    their two-stream schedule.  start() creates the events (outside the timed region); summary() needs an idle device."""

    def __init__(self, capacity, families=None):
        from ._lib import lib, check
        self.lib, self.check = lib, check
        self.capacity = int(capacity)
        self.mask = sum(1 << f for f in (families if families is not None else KT_FAMILIES))
        self.dropped = 0
        self._agg = None

    def start(self):
        self._agg = None
        self.check(self.lib.aide_ktimer_start(self.mask, self.capacity), 'ktimer_start')

    def stop(self):
        self.lib.aide_ktimer_stop()

    def arm(self):
        self.check(self.lib.aide_ktimer_arm(self.mask), 'ktimer_arm')

    def summary(self):
        """per family of the MFMA convolution kernels (cached: the library has ONE timer, another DispatchTimer's start()
        discards what this one recorded)"""
        import ctypes
        if self._agg is not None:
            return self._agg
        torch.cuda.synchronize()
        agg = collections.OrderedDict()
        dropped = 0
        for fam, (name, exec_frac) in KT_FAMILIES.items():
            n, ms, fl, mx = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            rc = self.lib.aide_ktimer_read(fam, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(mx))
            if rc < 0:
                raise RuntimeError('aide_amd: ktimer_read failed (%d)' % rc)
            dropped = rc
            if n.value == 0:
                continue
            a = dict(launches=n.value, ms=ms.value, flops=fl.value, executed=fl.value * exec_frac, max_ms=mx.value)
            a['avg_ms'] = a['ms'] / a['launches']
            a['tflops'] = a['flops'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] > 0 else 0.0
            a['executed_tflops'] = a['executed'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] > 0 else 0.0
            agg[name] = a
        self.dropped = dropped
        self._agg = agg
        return agg

    def timeline(self):
        """every recorded launch in launch order: dict(family id, name, work, t0, t1 in ms after the first recorded dispatch's
        begin, stream handle); needs an idle device"""
        import ctypes
        torch.cuda.synchronize()
        n = self.capacity
        fam, work = (ctypes.c_int * n)(), (ctypes.c_double * n)()
        t0, t1, st = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_uint64 * n)()
        got = self.lib.aide_ktimer_dump(n, fam, work, t0, t1, st)
        if got < 0:
            raise RuntimeError('aide_amd: ktimer_dump failed (%d)' % got)
        return [dict(family=fam[i], name=KT_ALL.get(fam[i], 'family %d' % fam[i]), work=work[i], t0=t0[i], t1=t1[i], stream=st[i])
                for i in range(got)]


def queue_report(rows, steps, stream_names, peak_tflops, hbm_tbs=8.0):
    """rows: DispatchTimer.timeline() of `steps` whole steps with EVERY family armed -> (critical, streaming) blocks of the
    bench line.  critical: the launch stream with the most dispatch time -- busy time per step, its families by time, the
    dominant one priced (MFMA families: algorithmic and executed fraction of the matrix peak; streaming families: fraction of
    the HBM peak).  streaming: BatchNorm / pooling / up-sampling over all streams: algorithmic bytes / dispatch time."""
    if not rows:
        return None, None
    by_stream = collections.OrderedDict()
    for r in rows:
        by_stream.setdefault(r['stream'], []).append(r)
    busy = {s: sum(r['t1'] - r['t0'] for r in rs) for s, rs in by_stream.items()}
    crit = max(busy, key=busy.get)

    def families(rs):
        fams = collections.OrderedDict()
        for r in rs:
            a = fams.setdefault(r['name'], dict(launches=0, ms=0.0, work=0.0, family=r['family']))
            a['launches'] += 1
            a['ms'] += r['t1'] - r['t0']
            a['work'] += r['work']
        return fams

    def priced(name, a):
        d = dict(kernel=name, launches_per_step=round(a['launches'] / steps, 2), ms_per_step=round(a['ms'] / steps, 4))
        if a['family'] in KT_FAMILIES and a['ms'] > 0:
            tf = a['work'] / (a['ms'] * 1e-3) / 1e12
            d.update(bound='mfma', achieved_tflops=round(tf, 2), frac=round(tf / peak_tflops, 4),
                     executed_frac=round(tf * KT_FAMILIES[a['family']][1] / peak_tflops, 4))
        elif a['family'] in KT_STREAMING and a['ms'] > 0:
            tb = a['work'] / (a['ms'] * 1e-3) / 1e12
            d.update(bound='hbm', gb_per_step=round(a['work'] / steps / 1e9, 3), effective_tb_s=round(tb, 3), frac=round(tb / hbm_tbs, 4))
        return d
    fams = families(by_stream[crit])
    order = sorted(fams.items(), key=lambda kv: -kv[1]['ms'])
    span = max(r['t1'] for r in rows) - min(r['t0'] for r in rows)
    critical = dict(stream=stream_names.get(crit, hex(crit)), busy_ms_per_step=round(busy[crit] / steps, 4),
                    instrumented_ms_per_step=round(span / steps, 4),
                    streams={stream_names.get(s, hex(s)): dict(busy_ms_per_step=round(b / steps, 4), launches_per_step=round(len(by_stream[s]) / steps, 1))
                             for s, b in busy.items()},
                    dominant=priced(*order[0]),
                    families=[dict(kernel=k, ms_per_step=round(a['ms'] / steps, 4), share=round(a['ms'] / busy[crit], 4),
                                   launches_per_step=round(a['launches'] / steps, 2)) for k, a in order])
    allf = families(rows)
    streaming = collections.OrderedDict()
    for fid, name in KT_STREAMING.items():
        if name in allf:
            d = priced(name, allf[name])
            d.pop('kernel')
            d['ms_per_step_on_critical_stream'] = round(fams[name]['ms'] / steps, 4) if name in fams else 0.0
            streaming[name] = d
    tot_b = sum(allf[n]['work'] for n in streaming)
    tot_ms = sum(allf[n]['ms'] for n in streaming)
    if tot_ms > 0:
        streaming['all'] = dict(gb_per_step=round(tot_b / steps / 1e9, 3), ms_per_step=round(tot_ms / steps, 4),
                                effective_tb_s=round(tot_b / (tot_ms * 1e-3) / 1e12, 3), frac=round(tot_b / (tot_ms * 1e-3) / 1e12 / hbm_tbs, 4),
                                peak_tb_s=hbm_tbs, bytes='algorithmic: every operand of a layer read once, its result written once')
    return critical, streaming
