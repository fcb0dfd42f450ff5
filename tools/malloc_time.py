"""Measurement aid: what hipMalloc / first touch / hipFree of one buffer cost by size (GB) on the box.

    python tools/malloc_time.py [sizes in GB ...]
"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def t(gb):
    p = C.c_void_p()
    t0=time.perf_counter(); r=hip.hipMalloc(C.byref(p), C.c_size_t(int(gb*(1<<30)))); t1=time.perf_counter()
    hip.hipMemset(p, 0, C.c_size_t(int(gb*(1<<30)))); hip.hipDeviceSynchronize(); t2=time.perf_counter()
    hip.hipFree(p); t3=time.perf_counter()
    print(f"{gb:6.1f} GB: malloc {1e3*(t1-t0):8.2f} ms  first memset {1e3*(t2-t1):8.2f} ms  free {1e3*(t3-t2):8.2f} ms (rc {r})", flush=True)
hip.hipSetDevice(0)
import sys
sizes = [float(x) for x in sys.argv[1:]] or [0.4, 2, 5, 20, 40, 40, 80]
for gb in sizes:
    t(gb)
