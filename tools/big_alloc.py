import sys, ctypes, time
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched
lib = _lib.load()
for n, gb in ((7, 24.0), (4, 18.0)):
    sizes = (ctypes.c_size_t * n)(*[int(gb * 1e9)] * n)
    ptrs = (ctypes.c_void_p * n)()
    t0 = time.time()
    st = lib.crt_hip_buffer_alloc_set(n, sizes, ptrs)
    print(n, gb, "status", st, "time", round(time.time() - t0, 2), batched.buffer_stats(), "free GB", torch.cuda.mem_get_info()[0] / 1e9, flush=True)
    if st == 0:
        buf = ctypes.create_string_buffer(4096)
        lib.crt_hip_buffer_describe(ptrs[0], buf, 4096); print(buf.value.decode()[:80])
        for q in ptrs: lib.crt_hip_buffer_free(q)
    lib.crt_hip_buffer_trim()
    print("after trim free GB", torch.cuda.mem_get_info()[0] / 1e9)
# a second large set while the first one's chunks sit in the pool (freed, not trimmed): the pool's memory must count as available
sizes = (ctypes.c_size_t * 7)(*[int(24e9)] * 7); ptrs = (ctypes.c_void_p * 7)()
assert lib.crt_hip_buffer_alloc_set(7, sizes, ptrs) == 0
for q in ptrs: lib.crt_hip_buffer_free(q)
print("pool after free", batched.buffer_stats(), "free GB", torch.cuda.mem_get_info()[0] / 1e9)
sizes = (ctypes.c_size_t * 4)(*[int(43.2e9)] * 4); ptrs = (ctypes.c_void_p * 4)()
st = lib.crt_hip_buffer_alloc_set(4, sizes, ptrs)
print("second set status", st, batched.buffer_stats(), "free GB", torch.cuda.mem_get_info()[0] / 1e9)
assert st == 0
