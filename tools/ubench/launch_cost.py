import sys, os, time, ctypes
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from aide_amd._lib import lib
dll = lib.load()
x = torch.zeros(1, 1, 4, 4, device='cuda')
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = ctypes.c_void_p(x.data_ptr())
args = [p, ctypes.c_int64(16), ctypes.c_int(1), ctypes.c_int(1), ctypes.c_int(4), ctypes.c_int(4), s]
f0 = lambda: dll.aide_wgrad_queue_pending(None)
f1 = dll.aide_fill_zero
for f, a, name in ((f0, [], 'ctypes call, no launch'), (f1, args, 'ctypes call + 1 launch (fill_zero)')):
    for _ in range(200): f(*a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 2000
    for _ in range(n): f(*a)
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    print('%-40s %.2f us' % (name, dt * 1e6))
ev = ctypes.c_void_p(); dll.aide_event_create(ctypes.byref(ev))
s2 = torch.cuda.Stream(); sp2 = ctypes.c_void_p(s2.cuda_stream)
t = time.perf_counter()
for _ in range(2000): dll.aide_stream_order(ev, s, sp2)
print('%-40s %.2f us' % ('stream_order (record + wait)', (time.perf_counter() - t) / 2000 * 1e6))
