#!/usr/bin/env python3
"""Do the crowded cells of a collapsing column find room for their hit masks in the plan pool?
Steps a config n steps, runs the stages up to the production density pass once more and counts the
particles whose audit count is -1 (crowded cell without masks: full walk in the force pass).
    python tools/mask_room.py C3 300"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import native, scene

cfg, nsteps = sys.argv[1], int(sys.argv[2])
sc = scene.make_config(cfg)
p = native.make_params(sc.grid_size, body_bounds=(sc.tank_min, sc.tank_max))
ctx = native.Context()
ctx.set_boundary(sc.bpos, sc.bparams)
ctx.set_colliders(sc.boxes)
ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
ctx.step_resident(p, nsteps)
ctx.hash_position(p)
ctx.radix_sort()
ctx.map_table()
ctx.reorder()
ctx.compute_density_and_pressure(p)
cnt = ctx.read(native.SPH_BUF_MASK_COUNTS)
ctx.synchronize()
print(f"{cfg} after {nsteps} steps: {len(cnt)} particles, {(cnt < 0).sum()} in crowded cells without masks, "
      f"hits per particle mean {cnt[cnt >= 0].mean():.1f} max {cnt.max()}")
