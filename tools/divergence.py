#!/usr/bin/env python3
"""Divergence of the shipped scene's trajectory: GPU (libsphfluid) against the oracle, next to the
oracle against itself with the initial positions perturbed by one ulp.  WCSPH with k = 512 is
chaotic: a rounding difference grows until it saturates at the size of the flow's own
fluctuations.  The second curve is that floor: no implementation whose rounding differs from the
reference's in the last bit can stay closer than it.  Writes profiles/r2_divergence_demo.txt.
usage: python tools/divergence.py [--no-gpu]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsimulation_amd import scene  # noqa: E402
from oracle import oracle as O  # noqa: E402

MARKS = (1, 2, 5, 10, 20, 50, 100)


def oracle_traj(sc, pos0):
    pos, vel, boxes = pos0.copy(), sc.vel.copy(), sc.boxes.copy()
    dt = float(scene.DT)
    out = {}
    for step in range(1, max(MARKS) + 1):
        vel, _, _, _ = O.sph_step(pos, vel, sc.inv_mass, sc.params, sc.bpos, sc.bparams,
                                  sc.grid_size, dt=dt)
        pos, vel, imp = O.integrate_particles(pos, vel, sc.inv_mass, sc.params, boxes, dt)
        boxes = O.integrate_bodies(boxes, imp, dt)
        if step in MARKS:
            out[step] = (pos.copy(), vel.copy())
    return out


def norms(a, b):
    e = np.abs(a.astype(np.float64) - b)
    return e.max() / np.abs(b).max(), np.median(e) / np.abs(b).max()


def main():
    O.build()
    sc = scene.demo_scene()
    ref = oracle_traj(sc, sc.pos)
    pert = oracle_traj(sc, np.nextafter(sc.pos, np.float32(np.inf)))
    gpu = None
    if "--no-gpu" not in sys.argv:
        from fluidsimulation_amd import native
        p = native.make_params(sc.grid_size)
        gpu = {}
        with native.Context(device=0) as ctx:
            ctx.set_boundary(sc.bpos, sc.bparams)
            ctx.set_colliders(sc.boxes)
            ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
            done = 0
            for step in MARKS:
                ctx.step_resident(p, step - done)
                done = step
                gpu[step] = ctx.download()
    lines = ["# shipped scene (4800 + 4000 particles, walls, BuoyancyObject), dt = 1/60",
             "# error = |a - oracle| / max|oracle|, positions (max, median) and velocities (max, median)",
             "# step   oracle(+1ulp) pos max/med   vel max/med        gpu pos max/med   vel max/med"]
    for step in MARKS:
        pp = norms(pert[step][0], ref[step][0])
        pv = norms(pert[step][1], ref[step][1])
        row = f"{step:5d}   {pp[0]:.2e} {pp[1]:.2e}   {pv[0]:.2e} {pv[1]:.2e}"
        if gpu is not None:
            gp = norms(gpu[step][0], ref[step][0])
            gv = norms(gpu[step][1], ref[step][1])
            row += f"        {gp[0]:.2e} {gp[1]:.2e}   {gv[0]:.2e} {gv[1]:.2e}"
        lines.append(row)
    text = "\n".join(lines) + "\n"
    print(text)
    out = os.path.join(ROOT, "gpurun_out" if gpu is not None else "/tmp", "r2_divergence_demo.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
