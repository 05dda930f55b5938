import torch, ctypes
print('priority_range', torch.cuda.Stream.priority_range())
s = torch.cuda.Stream(); print('pool stream priority', s.priority, 'default stream priority', torch.cuda.current_stream().priority, torch.cuda.default_stream().priority)
hip = ctypes.CDLL('libamdhip64.so')
p = ctypes.c_int(99)
for name, h in (('null', 0), ('pool', s.cuda_stream)):
    rc = hip.hipStreamGetPriority(ctypes.c_void_p(h), ctypes.byref(p)); print(name, 'hipStreamGetPriority rc', rc, 'prio', p.value)
lo, hi = ctypes.c_int(), ctypes.c_int()
print('range rc', hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), lo.value, hi.value)
for pr in (-1, 0, 1):
    t = torch.cuda.Stream(priority=pr); hip.hipStreamGetPriority(ctypes.c_void_p(t.cuda_stream), ctypes.byref(p)); print('torch priority', pr, '-> torch says', t.priority, 'hip says', p.value)
