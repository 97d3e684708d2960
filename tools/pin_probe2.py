#!/usr/bin/env python3
"""hipHostRegister on this box, straight through the runtime: which pointers / sizes / flags work?"""
import ctypes as C, mmap
import numpy as np
import torch  # loads torch's HIP runtime first
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipGetErrorString.restype = C.c_char_p
hip.hipGetErrorString.argtypes = [C.c_int]
print("hipInit/dev", hip.hipSetDevice(0))
def tryreg(name, ptr, size, flags):
    r = hip.hipHostRegister(ptr, size, flags)
    print(f"{name}: ptr%4096={ptr % 4096} size={size} flags={flags} -> {r} {hip.hipGetErrorString(r).decode()}")
    if r == 0:
        hip.hipHostUnregister(C.c_void_p(ptr))
    hip.hipGetLastError()
a = np.empty(64 << 20, np.uint8)
tryreg("numpy 64MB", a.ctypes.data, a.nbytes, 0)
off = (-a.ctypes.data) % 4096
tryreg("numpy aligned", a.ctypes.data + off, (a.nbytes - off) // 4096 * 4096, 0)
for flags in (1, 2, 4, 8):
    tryreg("numpy aligned flags", a.ctypes.data + off, (a.nbytes - off) // 4096 * 4096, flags)
m = mmap.mmap(-1, 64 << 20)
buf = (C.c_char * (64 << 20)).from_buffer(m)
m[:] = b"\0" * (64 << 20)
tryreg("mmap anonymous", C.addressof(buf), 64 << 20, 0)
small = np.empty(1 << 20, np.uint8)
tryreg("numpy 1MB", small.ctypes.data, small.nbytes, 0)
p = C.c_void_p()
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
print("hipHostMalloc 64MB ->", hip.hipHostMalloc(C.byref(p), 64 << 20, 0))
