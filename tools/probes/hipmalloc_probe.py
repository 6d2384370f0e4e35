"""hipMalloc / hipFree wall time against size on THIS box (no caching allocator in between): the cold build's large work
buffers are first allocations, and their cost differs between boxes of the pool by two orders of magnitude.
    python tools/probes/hipmalloc_probe.py"""
import ctypes as C, time, platform
hip = C.CDLL('libamdhip64.so')
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
assert hip.hipSetDevice(0) == 0
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), 1 << 20) == 0          # runtime start-up is not what is measured
hip.hipFree(p)
print('kernel', platform.release())
for rep in range(2):
    for mb in (16, 64, 128, 256, 384, 512, 768, 1024, 2048, 4096):
        t0 = time.perf_counter()
        rc = hip.hipMalloc(C.byref(p), mb << 20)
        t1 = time.perf_counter()
        hip.hipMemset(p, 0, 4096)
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        hip.hipFree(p)
        t3 = time.perf_counter()
        print('round %d: %5d MB  hipMalloc %8.3f ms  first touch %7.3f ms  hipFree %8.3f ms  rc %d' % (rep, mb, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), rc))
# several buffers alive at once (a build holds the matrix, its image and the work buffers together)
held = []
t0 = time.perf_counter()
for i in range(8):
    q = C.c_void_p()
    t1 = time.perf_counter()
    hip.hipMalloc(C.byref(q), 128 << 20)
    held.append(q)
    print('128 MB no. %d while the others are held: %.3f ms' % (i, 1e3 * (time.perf_counter() - t1)))
for q in held:
    hip.hipFree(q)
# cumulative: does the cost appear at a total, not at a size?
held = []
total = 0
for mb in [76, 76, 1, 76, 76, 471, 400, 100, 100, 512, 512, 1024, 1024, 2048, 2048]:
    q = C.c_void_p()
    t1 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(q), mb << 20)
    dt = 1e3 * (time.perf_counter() - t1)
    hip.hipMemset(q, 0, 4096)
    hip.hipDeviceSynchronize()
    held.append(q)
    total += mb
    print('%5d MB on top of %5d MB held: %8.3f ms  rc %d' % (mb, total - mb, dt, rc))
for q in held:
    hip.hipFree(q)
