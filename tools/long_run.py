#!/usr/bin/env python3
"""Stability of long resident runs: python tools/long_run.py C3 600 [contacts] -- finite state, particles
inside the tank, no sticky plan error (raised by synchronize/download), occupancy extremes."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import native, scene

cfg, nsteps = sys.argv[1], int(sys.argv[2])
contacts = len(sys.argv) > 3 and sys.argv[3] == "contacts"
sc = scene.make_config(cfg)
p = native.make_params(sc.grid_size, body_bounds=(sc.tank_min, sc.tank_max), body_contacts=contacts)
ctx = native.Context()
ctx.set_boundary(sc.bpos, sc.bparams)
ctx.set_colliders(sc.boxes)
ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
done = 0
while done < nsteps:
    k = min(100, nsteps - done)
    t0 = time.perf_counter()
    ctx.step_resident(p, k)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / k
    done += k
    pos, vel = ctx.download()
    boxes = ctx.get_colliders()
    inside = bool(np.all(pos >= sc.tank_min - 0.5) and np.all(pos <= sc.tank_max + 0.5))
    print(f"{cfg} step {done}: {dt * 1e3:.3f} ms/step finite {bool(np.isfinite(pos).all() and np.isfinite(vel).all())} "
          f"inside {inside} |v|max {np.abs(vel).max():.2f} boxes finite {bool(np.isfinite(boxes).all())}", flush=True)
