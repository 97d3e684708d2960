#!/usr/bin/env python3
"""Integrate-stage time of C5 (4.19 M particles) against the number of box colliders: the 6 walls +
64 dynamic boxes of the configuration, then the same plus 1000 and 4000 small static obstacles in a
lattice through the tank.  usage: python tools/collider_scaling.py > profiles/r2_collider_scaling.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsimulation_amd import native, scene

sc = scene.make_config("C5")
p = native.make_params(sc.grid_size, body_bounds=(sc.tank_min, sc.tank_max))
rng = np.random.default_rng(5)
L = sc.tank_max - sc.tank_min
base = sc.boxes.copy()
print("# C5: %d fluid particles; integrate = ApplyForce + integrate + colliders (k_apply_integrate)" % sc.n)
print("# boxes   ms/step   integrate ms   bodies ms")
for extra in (0, 1000, 4000):
    m = int(round(extra ** (1 / 3))) if extra else 0
    obs = []
    for k in range(extra):
        ix, iy, iz = k % m, (k // m) % m, k // (m * m)
        c = sc.tank_min + (np.array([ix, iy, iz]) + 0.5) / m * L + rng.uniform(-0.1, 0.1, 3)
        obs.append(scene.static_box(c, rng.uniform(0.05, 0.2, 3), restitution=0.3))
    boxes = np.concatenate([base, np.stack(obs).astype(np.float32)]) if obs else base
    with native.Context(device=0, timings=True) as ctx:
        ctx.set_boundary(sc.bpos, sc.bparams)
        ctx.set_colliders(boxes)
        ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
        ctx.step_resident(p, 10)
        ctx.synchronize()
        ctx.reset_timings()
        import time
        t0 = time.perf_counter()
        ctx.step_resident(p, 30)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 30 * 1e3
        tm = ctx.timings()
        print(f"{len(boxes):6d} {dt:9.3f} {tm['ms']['integrate'] / 30:12.3f} {tm['ms']['bodies'] / 30:10.3f}", flush=True)
