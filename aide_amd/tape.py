"""Launch tapes: re-issue the launch sequence of a forward / backward pass without the Python-side marshalling.

The whole-network forward and backward are flat sequences of C-ABI calls whose arguments are the same every step except
for the addresses of a few per-call tensors (network inputs, logits, incoming gradient, the flat gradient arena).  The
first pass over a plan records every call as (ctypes function, ctypes argument objects, return code); later passes patch
the addresses of the per-call tensors in place and call the functions again: ~1 us of Python per launch instead of ~15
(tensor-layout checks, stride arithmetic, ctypes conversion).  Python callbacks that must run between launches (the
data-parallel bucket hooks) are tape entries as well.

A tape is only valid for the exact configuration it was recorded under; the engine keys it on the launch stream, the
shapes / strides of the per-call tensors and the addresses of every parameter and buffer of the module."""
import ctypes

from . import _lib


class Tape(object):
    def __init__(self, key):
        self.key = key
        self.calls = []          # (cfn, [ctypes args], rc, name) | (None, callable, tag, 'py')
        self.patches = None      # [(c_void_p object, index of the dynamic tensor, byte offset)]
        self.bases = None

    # ---- recording
    def __enter__(self):
        assert _lib.TAPE[0] is None, 'nested tape recording'
        _lib.TAPE[0] = self
        return self

    def __exit__(self, et, ev, tb):
        _lib.TAPE[0] = None
        return False

    def py(self, fn, tag=None):
        """record a Python callback (runs at this position of every replay) -- the caller also runs it now"""
        self.calls.append((None, fn, tag, 'py'))

    def finish(self, dynamic):
        """dynamic: the per-call tensors, in the order replay() will receive them."""
        rng = [(t.data_ptr(), t.data_ptr() + max(t.numel(), 1) * t.element_size() if t.is_contiguous()
                else t.data_ptr() + _span_bytes(t)) for t in dynamic]
        self.patches = []
        for cfn, cargs, rc, name in self.calls:
            if cfn is None:
                continue
            for a in cargs:
                if isinstance(a, ctypes.c_void_p) and a.value:
                    v = a.value
                    for j, (lo, hi) in enumerate(rng):
                        if lo <= v < hi:
                            self.patches.append((a, j, v - lo))
                            break
        self.bases = [lo for lo, _ in rng]
        return self

    # ---- replay
    def replay(self, dynamic, skip_tags=()):
        bases = [t.data_ptr() for t in dynamic]
        if bases != self.bases:
            for a, j, off in self.patches:
                a.value = bases[j] + off
            self.bases = bases
        for cfn, cargs, rc, name in self.calls:
            if cfn is None:
                if rc in skip_tags:
                    continue
                cargs()
            elif name in skip_tags:
                continue
            else:
                got = cfn(*cargs)
                if got != rc:
                    raise RuntimeError('aide_amd: %s returned %d on replay (recorded %d)' % (name, got, rc))


def _span_bytes(t):
    """bytes from data_ptr() to one past the last element of a strided tensor"""
    span = 1
    for n, s in zip(t.shape, t.stride()):
        if n > 1:
            span += (n - 1) * s
    return span * t.element_size()
