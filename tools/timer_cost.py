#!/usr/bin/env python3
"""What the per-stage HIP events cost: C3, 50 + 100 resident steps with SphConfig.enableTimings on and off."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import native, scene

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = scene.make_config(cfg)
p = native.make_params(sc.grid_size, body_bounds=(sc.tank_min, sc.tank_max))
for timings in (True, False, True, False):
    ctx = native.Context(timings=timings)
    ctx.set_boundary(sc.bpos, sc.bparams)
    ctx.set_colliders(sc.boxes)
    ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
    ctx.step_resident(p, 50)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.step_resident(p, 100)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print(f"{cfg} timings={timings}: {dt * 1e3:.3f} ms/step  {sc.n / dt:.4g} updates/s", flush=True)
    del ctx
