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
