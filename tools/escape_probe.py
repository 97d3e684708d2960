#!/usr/bin/env python3
"""Where does the first particle leave the tank?  python tools/escape_probe.py C5 600 [contacts]
Steps 10 at a time; at the first escape goes back to the state 10 steps before (re-upload) and
steps one at a time, printing the particle and the boxes near it."""
import sys, os
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
lo, hi = sc.tank_min + 0.25 - 1e-3, sc.tank_max - 0.25 + 1e-3   # centres stay a radius inside
prev = (sc.pos.copy(), sc.vel.copy(), sc.boxes.copy())
done = 0
while done < nsteps:
    ctx.step_resident(p, 10)
    done += 10
    pos, vel = ctx.download()
    boxes = ctx.get_colliders()
    out = np.flatnonzero(np.any((pos < lo) | (pos > hi), axis=1))
    if len(out) == 0:
        prev = (pos, vel, boxes)
        continue
    i = int(out[0])
    print(f"step {done}: {len(out)} particles outside; following particle {i}")
    ctx.set_colliders(prev[2])
    ctx.upload(prev[0], prev[1], sc.inv_mass, sc.params, sc.grid_size)
    for s in range(done - 10, done):
        pos0, vel0 = ctx.download()
        b0 = ctx.get_colliders()
        ctx.step_resident(p, 1)
        pos1, vel1 = ctx.download()
        d = np.linalg.norm(b0[:, 0:3] - pos0[i], axis=1) - np.linalg.norm(b0[:, 7:10], axis=1)
        near = np.argsort(d)[:3]
        inside = bool(np.all(pos1[i] >= lo) and np.all(pos1[i] <= hi))
        print(f" step {s}->{s+1}: x {pos0[i]} v {vel0[i]} |v|dt {np.linalg.norm(vel0[i]) / 60:.3f} -> x {pos1[i]} v {vel1[i]} inside {inside}")
        for k in near:
            print(f"    box {k}: c {b0[k, 0:3]} he {b0[k, 7:10]} lv {b0[k, 10:13]} av {b0[k, 13:16]} dyn {b0[k, 16] != 0}")
        if not inside:
            np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "escape_state.npz"),
                     i=i, pos0=pos0[i], vel0=vel0[i], pos1=pos1[i], vel1=vel1[i], boxes=b0,
                     near_pos=pos0[np.linalg.norm(pos0 - pos0[i], axis=1) < 3.0],
                     near_vel=vel0[np.linalg.norm(pos0 - pos0[i], axis=1) < 3.0])
            break
    break
else:
    print("no particle left the tank")
