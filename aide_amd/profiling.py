"""Per-kernel timing with HIP events recorded on the launch stream (used by bench.py's roofline)."""
import collections

import torch


class KernelTimer(object):
    """begin(tag, flops, executed) / end() bracket ONE kernel-family launch with a HIP event pair on the
    current stream. `flops` is the algorithmic (direct-convolution) count, `executed` what the MFMA pipe
    really multiplies (16/36 of it for a Winograd F(2x2,3x3) launch). Events are resolved after a device
    synchronise by summary()."""

    def __init__(self):
        self.records = []
        self._open = None

    def begin(self, tag, flops, executed=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._open = (tag, flops, flops if executed is None else executed, e0)

    def end(self):
        tag, flops, executed, e0 = self._open
        e1 = torch.cuda.Event(enable_timing=True)
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
