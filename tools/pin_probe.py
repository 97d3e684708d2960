#!/usr/bin/env python3
"""Why does SphConfig.pinHostArrays change nothing on this box?  Registers a 200 MB array and prints
what the driver says, the process's locked-memory limit, and the copy rates either way."""
import os, sys, time, resource
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import native, scene

print("RLIMIT_MEMLOCK", resource.getrlimit(resource.RLIMIT_MEMLOCK))
sc = scene.make_config("C2")
n = 16 * sc.n
pos = np.tile(sc.pos, (16, 1)); vel = np.zeros_like(pos)
inv = np.tile(sc.inv_mass, 16); par = np.tile(sc.params, (16, 1))
for pin in (False, True):
    with native.Context(pin_host_arrays=pin) as c:
        c.n = n
        for _ in range(3):
            t0 = time.perf_counter()
            c.upload(pos, vel, inv, par, sc.grid_size)
            t1 = time.perf_counter()
            c.download()
            t2 = time.perf_counter()
        print("pin", pin, c.pin_stats(), c.lib.sph_last_error(c.h).decode())
        print(f"   upload {24 * n / (t1 - t0) / 1e9:.1f} GB/s  download {24 * n / (t2 - t1) / 1e9:.1f} GB/s (host time incl. pack / material scan)")
