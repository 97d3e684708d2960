#!/usr/bin/env python3
"""Times the parts of the host round trip (sph_upload, sph_step_resident, sph_download) on C3,
pageable and registered host arrays.  usage: python tools/round_trip_parts.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsimulation_amd import native, scene
import ctypes as C

sc = scene.make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
p = native.make_params(sc.grid_size)
for pin in (False, True):
    with native.Context(device=0, pin_host_arrays=pin) as ctx:
        ctx.set_boundary(sc.bpos, sc.bparams)
        pos, vel = sc.pos.copy(), sc.vel.copy()
        lib, h = ctx.lib, ctx.h
        def up():
            ctx._chk(lib.sph_upload(h, native._ptr(pos), native._ptr(vel), native._ptr(sc.inv_mass), native._ptr(sc.params), sc.n, sc.grid_size))
        def down():
            ctx._chk(lib.sph_download(h, native._ptr(pos), native._ptr(vel)))
        ctx.n = sc.n
        up(); ctx.step_resident(p, 1); ctx.synchronize(); down()
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); up(); t1 = time.perf_counter()
            ctx.step_resident(p, 1); ctx.synchronize(); t2 = time.perf_counter()
            down(); t3 = time.perf_counter()
            t.append((t1 - t0, t2 - t1, t3 - t2))
        t = np.array(t).min(axis=0) * 1e3
        print(f"pinned={pin}: upload {t[0]:.1f} ms  step {t[1]:.1f} ms  download {t[2]:.1f} ms  ({sc.n} particles, 24 B each way)")
