"""Per-kernel timing with HIP events recorded on the launch stream (used by bench.py's roofline)."""
import collections

import torch


class KernelTimer(object):
    """begin(tag, flops) / end() bracket ONE kernel-family launch with a HIP event pair on the
    current stream. Events are resolved after a device synchronise by summary()."""

    def __init__(self):
        self.records = []
        self._open = None

    def begin(self, tag, flops):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._open = (tag, flops, e0)

    def end(self):
        tag, flops, e0 = self._open
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((tag, flops, e0, e1))
        self._open = None

    def summary(self):
        torch.cuda.synchronize()
        agg = collections.OrderedDict()
        for tag, flops, e0, e1 in self.records:
            a = agg.setdefault(tag, dict(launches=0, ms=0.0, flops=0.0))
            a['launches'] += 1
            a['ms'] += e0.elapsed_time(e1)
            a['flops'] += flops
        for a in agg.values():
            a['avg_ms'] = a['ms'] / max(a['launches'], 1)
            a['tflops'] = a['flops'] / (a['ms'] * 1e-3) / 1e12 if a['ms'] > 0 else 0.0
        return agg
