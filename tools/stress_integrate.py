#!/usr/bin/env python3
"""Row I under extreme speeds: sph_integrate against the oracle for particles at |v| dt up to ~10
cell sizes among walls and dynamic boxes; reports mismatches and escapes in both."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import native, scene
from oracle import oracle
from tests.test_gpu_steps import _many_body_scene

for std in (40.0, 150.0, 400.0):
    sc = _many_body_scene(21)
    rng = np.random.default_rng(int(std))
    sc.vel = rng.normal(0, std, sc.pos.shape).astype(np.float32)
    p = native.make_params(sc.grid_size)
    with native.Context() as ctx:
        ctx.set_boundary(sc.bpos, sc.bparams)
        ctx.set_colliders(sc.boxes)
        ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
        ctx.integrate(p)
        gpos, gvel = ctx.download()
    opos, ovel, _ = oracle.integrate_particles(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.boxes, float(p.dt))
    lo, hi = sc.tank_min + 0.25 - 1e-3, sc.tank_max - 0.25 + 1e-3
    gout = np.any((gpos < lo) | (gpos > hi), axis=1)
    oout = np.any((opos < lo) | (opos > hi), axis=1)
    err = np.abs(gpos - opos).max(axis=1)
    bad = np.flatnonzero(err > 1e-3)
    print(f"std {std}: n {sc.n} |v|dt max {np.linalg.norm(sc.vel, axis=1).max() / 60:.2f}; gpu outside {gout.sum()} oracle outside {oout.sum()}; "
          f"positions differing by > 1e-3: {len(bad)}")
    for i in bad[:5]:
        print("   ", i, sc.pos[i], sc.vel[i], "gpu", gpos[i], gvel[i], "oracle", opos[i], ovel[i])
    for i in np.flatnonzero(oout)[:3]:
        print("    oracle outside:", i, sc.pos[i], sc.vel[i], "->", opos[i], ovel[i])
