"""ctypes binding of libaide_hip.so. Prototypes are parsed from include/aide_hip.h so the header is
the single source of truth for the C ABI. There is NO fallback: if the library is missing the
import of any compute entry point fails loudly."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, 'include', 'aide_hip.h')
LIB_PATH = os.environ.get('AIDE_HIP_LIB', os.path.join(HERE, 'libaide_hip.so'))   # override: probe builds

_SCALARS = {
    'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'size_t': ctypes.c_size_t,
    'aide_stream_t': ctypes.c_void_p, 'double': ctypes.c_double,
}


def _ctype(decl):
    decl = re.sub(r'/\*.*?\*/', '', decl).strip()
    if '*' in decl:
        return ctypes.c_void_p
    toks = [t for t in decl.split() if t != 'const']
    ty = toks[0] if len(toks) <= 2 else ' '.join(toks[:-1])
    if ty not in _SCALARS:
        raise ValueError('unknown C type in header: %r' % decl)
    return _SCALARS[ty]


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'\b(int|size_t)\s+(aide_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argl = [a.strip() for a in args.split(',')] if args.strip() and args.strip() != 'void' else []
        protos[name] = (_SCALARS[ret], [_ctype(a) for a in argl])
    return protos


TAPE = [None]          # the Tape that records the calls of the running forward / backward (aide_amd/tape.py), or None
# AIDE_ABI_COVERAGE=<file>: which entry points of the C ABI a process reaches (tools/abi_coverage.py merges the files of a test /
# bench run into profiles/rNN_abi_coverage.md); a measurement aid, off by default
COVER = {} if os.environ.get('AIDE_ABI_COVERAGE') else None
if COVER is not None:
    import atexit
    import json

    def _dump_cover():
        path = '%s.%d' % (os.environ['AIDE_ABI_COVERAGE'], os.getpid())
        with open(path, 'w') as f:
            json.dump(COVER, f)
    atexit.register(_dump_cover)


class _Fn(object):
    """One entry point of the C ABI.  Calls go straight to ctypes; while a Tape is recording they are converted to
    ctypes argument objects once and appended to it, so that later steps can re-issue the same launch sequence without
    any of the Python-side marshalling (aide_amd/tape.py)."""
    __slots__ = ('cfn', 'argtypes', 'name')

    def __init__(self, cfn, argtypes, name):
        self.cfn, self.argtypes, self.name = cfn, argtypes, name

    def __call__(self, *args):
        if COVER is not None:
            COVER[self.name] = COVER.get(self.name, 0) + 1
        tape = TAPE[0]
        if tape is None:
            return self.cfn(*args)
        cargs = []
        for a, t in zip(args, self.argtypes):
            if t is ctypes.c_void_p:
                # pointers: int / None / c_void_p / byref()-style objects.  A fresh c_void_p per argument: the tape patches
                # the addresses of per-call tensors in place
                if a is None or isinstance(a, int):
                    a = ctypes.c_void_p(a)
                elif isinstance(a, ctypes.c_void_p):
                    a = ctypes.c_void_p(a.value)
                # anything else (arrays, byref): keep the object (and with it its referent) alive as is
            elif not isinstance(a, t):
                a = t(a)
            cargs.append(a)
        if len(cargs) != len(self.argtypes):
            raise TypeError('%s: %d arguments, %d expected' % (self.name, len(cargs), len(self.argtypes)))
        rc = self.cfn(*cargs)
        tape.calls.append((self.cfn, cargs, rc, self.name))
        return rc


class _Lib(object):
    def __init__(self):
        self._dll = None
        self._fns = {}
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    'aide_amd: %s is missing. Build it with `python -m aide_amd.build` '
                    '(hipcc, gfx950). There is no CPU/eager fallback.' % LIB_PATH)
            # torch must map ITS libamdhip64 first: the kernels launch on torch's streams, so both
            # have to share one HIP runtime (the dynamic linker then resolves our DT_NEEDED entry to
            # the already-loaded library of the same SONAME instead of /opt/rocm's copy)
            import torch  # noqa: F401
            dll = ctypes.CDLL(LIB_PATH)
            for name, (ret, args) in self.protos.items():
                try:
                    fn = getattr(dll, name)      # AttributeError if the symbol is not exported
                except AttributeError:
                    if 'AIDE_HIP_LIB' in os.environ:     # an A-B build of an older tree (tools/probes/mk_probe.py): fails when called
                        continue
                    raise
                fn.restype = ret
                fn.argtypes = args
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith('aide_'):
            fn = self._fns.get(name)
            if fn is None:
                cfn = getattr(self.load(), name)
                fn = self._fns[name] = _Fn(cfn, self.protos[name][1], name)
            return fn
        raise AttributeError(name)


lib = _Lib()


def check(rc, what):
    if rc != 0:
        raise RuntimeError('aide_amd: %s failed with code %d (%s)' % (
            what, rc, 'bad argument' if rc < 0 else 'hipError'))
