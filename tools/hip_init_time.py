#!/usr/bin/env python3
"""Where the first context's 0.2 s go: the HIP runtime's start-up, call by call (ctypes on the runtime libtcmi.so binds to)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.perf_counter()
from trueconsense_amd import _ffi
lib = _ffi.lib()
t1 = time.perf_counter()
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = C.CDLL(line.split()[-1]); break
def T(name, fn):
    s = time.perf_counter(); r = fn(); print("  %-28s %7.1f ms (rc %s)" % (name, 1e3 * (time.perf_counter() - s), r))
print("import + dlopen libtcmi.so   %7.1f ms" % (1e3 * (t1 - t0)))
n = C.c_int(0)
T("hipInit", lambda: hip.hipInit(0))
T("hipGetDeviceCount", lambda: hip.hipGetDeviceCount(C.byref(n)))
T("hipSetDevice", lambda: hip.hipSetDevice(0))
buf = C.create_string_buffer(4096)
T("hipGetDeviceProperties", lambda: hip.hipGetDevicePropertiesR0600(buf, 0) if hasattr(hip, "hipGetDevicePropertiesR0600") else hip.hipGetDeviceProperties(buf, 0))
s = C.c_void_p()
T("hipStreamCreateWithFlags", lambda: hip.hipStreamCreateWithFlags(C.byref(s), 1))
p = C.c_void_p()
T("hipMalloc 64 MB", lambda: hip.hipMalloc(C.byref(p), 64 << 20))
T("hipMemsetAsync + sync", lambda: (hip.hipMemsetAsync(p, 0, 64 << 20, s), hip.hipStreamSynchronize(s))[1])
h = C.c_void_p()
T("hipHostMalloc 16 MB", lambda: hip.hipHostMalloc(C.byref(h), 16 << 20, 0))
T("tcmi_ctx_create (2nd ctx)", lambda: lib.tcmi_ctx_create(0, C.byref(C.c_void_p())))
