#!/usr/bin/env python3
"""Does the collapse of a tall column look the same in the oracle and in the HIP path?  A 16 x 16 x 256
particle column (128 h tall, as C3's) in a small tank, 400 steps, statistics every 50 steps (the
trajectories are chaotic long before that: only distributions can agree).  GPU box:
    python tools/tall_column.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import native, scene
from oracle import oracle

nx, ny, nz = 16, 256, 16          # y is up in the reference's scenes
sc = scene.dam_break(nx, ny, nz, jitter_seed=123)
dt = float(scene.DT)
p = native.make_params(sc.grid_size)
ctx = native.Context()
ctx.set_boundary(sc.bpos, sc.bparams)
ctx.set_colliders(sc.boxes)
ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
opos, ovel = sc.pos.copy(), sc.vel.copy()
boxes = sc.boxes.copy()


def stats(pos, vel):
    sp = np.linalg.norm(vel, axis=1)
    out = np.any((pos < sc.tank_min - 0.5) | (pos > sc.tank_max + 0.5), axis=1).mean()
    up = 1 if sc.tank_max[1] - sc.tank_min[1] >= sc.tank_max[2] - sc.tank_min[2] else 2
    return (f"|v| mean {sp.mean():7.3f} p99 {np.percentile(sp, 99):7.2f} max {sp.max():8.2f}  "
            f"height mean {pos[:, up].mean():7.3f}  outside {100 * out:6.3f} %")


print(f"{sc.n} particles, tank {sc.tank_min} .. {sc.tank_max}", flush=True)
for chunk in range(8):
    ctx.step_resident(p, 50)
    gpos, gvel = ctx.download()
    for _ in range(50):
        ovel, dens, pres, forces = oracle.sph_step(opos, ovel, sc.inv_mass, sc.params, sc.bpos,
                                                   sc.bparams, sc.grid_size, dt=dt)
        opos, ovel, imp = oracle.integrate_particles(opos, ovel, sc.inv_mass, sc.params, boxes, dt)
    print(f"step {50 * (chunk + 1):3d}  hip    {stats(gpos, gvel)}", flush=True)
    print(f"          oracle {stats(opos, ovel)}", flush=True)
