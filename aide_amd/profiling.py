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

    def start(self):
        self.check(self.lib.aide_ktimer_start(self.mask, self.capacity), 'ktimer_start')

    def stop(self):
        self.lib.aide_ktimer_stop()

    def arm(self):
        self.check(self.lib.aide_ktimer_arm(self.mask), 'ktimer_arm')

    def summary(self):
        import ctypes
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
        return agg
