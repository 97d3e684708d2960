#!/usr/bin/env python3
"""What does slab mode cost a rank per step beside the transfer itself?  C3 on one GPU, plain
resident loop against slab mode with a one-rank communicator (count pass, host synchronisation,
pack pass, k_slab_keys instead of the keys written by the integrate kernel, one step per call).
    python tools/slab_overhead.py [config] [steps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import native, scene, slab

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc = scene.make_config(cfg, jittered=True)
p = native.make_params(sc.grid_size, body_bounds=(sc.tank_min, sc.tank_max))


def timed(run, ctx):
    run(5)
    ctx.synchronize()
    ctx.reset_timings()
    t0 = time.perf_counter()
    run(steps)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    tm = ctx.timings()
    return dt, {k: round(v / max(tm["steps"], 1), 3) for k, v in tm["ms"].items()}


with native.Context(timings=1) as ctx:
    ctx.set_boundary(sc.bpos, sc.bparams)
    ctx.set_colliders(sc.boxes)
    ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
    dt, st = timed(lambda n: ctx.step_resident(p, n), ctx)
    print(f"{cfg} plain resident loop: {dt:.3f} ms/step {st}")

with native.Context(timings=1) as ctx:
    table, index = slab.material_table(sc.inv_mass, sc.params)
    ctx.set_boundary(sc.bpos, sc.bparams)
    ctx.set_colliders(sc.boxes)
    ids = np.arange(sc.n, dtype=np.uint32)
    ctx.slab_upload(sc.pos, sc.vel, index, ids, table, sc.grid_size, capacity=sc.n + 1024)
    ctx.slab_comm_init(0, 1, native.Context.slab_comm_unique_id())
    ctx.slab_exchange_reserve(1 << 20)
    # cuts inside the fluid on both sides, as an interior rank has them: the pack kernels find
    # records for both neighbours (there is nobody to send them to)
    xs = sc.pos[:, 0]
    xlo, xhi = float(np.percentile(xs, 1)), float(np.percentile(xs, 99))

    def run(n):
        for _ in range(n):
            ctx.slab_exchange(-np.inf, np.inf, 2.0 * sc.grid_size)
            ctx.step_resident(p, 1)
    dt2, st2 = timed(run, ctx)
    print(f"{cfg} slab mode, one rank:  {dt2:.3f} ms/step {st2}")
    print(f"overhead {dt2 - dt:.3f} ms/step ({100 * (dt2 / dt - 1):.1f} %)")
