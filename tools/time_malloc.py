"""hipMalloc / hipFree cost on this box (the first step of a mesh allocates ~80 GB of tables and scratch)."""
import ctypes as C
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize()
for gb in (1, 8, 34, 34):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(gb << 30))
    t1 = time.perf_counter()
    hip.hipMemset(p, 0, C.c_size_t(gb << 30)); hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipFree(p)
    t3 = time.perf_counter()
    print(f"{gb:3d} GB: hipMalloc {1e3 * (t1 - t0):8.1f} ms (rc {rc}), first touch (memset) {1e3 * (t2 - t1):8.1f} ms, hipFree {1e3 * (t3 - t2):8.1f} ms")
