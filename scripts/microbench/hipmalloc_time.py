"""Times hipMalloc / hipFree by size (what a table's column buffers cost to obtain)."""
import ctypes, time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
assert hip.hipSetDevice(0) == 0
p = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(p), 1 << 20) == 0
hip.hipFree(p)
for size in (4 << 10, 1 << 20, 8 << 20, 16 << 20, 64 << 20, 256 << 20, 1 << 30):
    ptrs, t0 = [], time.perf_counter()
    for _ in range(16):
        q = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(q), size) == 0
        ptrs.append(q)
    t1 = time.perf_counter()
    for q in ptrs:
        hip.hipFree(q)
    t2 = time.perf_counter()
    print(f"{size >> 10:>8} KiB  malloc {1e6 * (t1 - t0) / 16:8.1f} us  free {1e6 * (t2 - t1) / 16:8.1f} us", flush=True)
