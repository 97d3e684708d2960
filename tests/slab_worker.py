"""One rank of a slab-decomposition test; launched by tests/test_slab.py under
torch.distributed.run (gloo).  argv: backend ("numpy" | "hip") nsteps out_json bodies(0|1)

Every rank builds the same small dam-break scene, keeps its slab, runs nsteps with
SlabSimulation and sends its owned particles to rank 0, which compares them, by particle id,
with the undecomposed run of the same backend family (oracle for "numpy", one libsphfluid
context for "hip") and writes the verdict."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fluidsimulation_amd import scene, slab  # noqa: E402


def build_scene(bodies):
    sc = scene.dam_break(16, 10, 10, jitter_seed=0x51AB)
    if bodies:
        # a rigid box dipped into the fluid across the future slab face: both ranks push it
        box = scene.dynamic_box((4.0, 4.6, 2.5), (1.0, 0.5, 1.0), 2000.0)
        sc.boxes = np.concatenate([sc.boxes, box[None, :]]).astype(np.float32)
    # a shear across the future slab face so that particles change owner within a few steps
    mid = 0.5 * (sc.pos[:, 0].min() + sc.pos[:, 0].max())
    sc.vel[:, 0] = np.where(sc.pos[:, 2] > np.median(sc.pos[:, 2]), 6.0, -6.0).astype(np.float32)
    return sc, float(mid)


def main():
    backend_name, nsteps, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    bodies = len(sys.argv) > 4 and sys.argv[4] == "1"
    rebalance = len(sys.argv) > 5 and sys.argv[5] == "1"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sc, mid = build_scene(bodies)
    h = sc.grid_size
    span = sc.pos[:, 0].max() - sc.pos[:, 0].min()
    edges = [-np.inf] + [float(sc.pos[:, 0].min() + span * r / world) for r in range(1, world)] \
        + [np.inf]
    if rebalance:   # start from badly placed cuts: most of the fluid on rank 0
        x0 = float(sc.pos[:, 0].min())
        edges = [-np.inf] + [x0 + 0.60 * span + 2.2 * h * (r - 1) for r in range(1, world)] + [np.inf]
    # boundary particles within 3 h of the slab (they enter the densities of the ghosts too); with
    # moving cuts every rank keeps the whole boundary
    own, bsel = slab.split_scene(sc, edges, rank, margin=(1e9 if rebalance else 3.0 * h))
    table, index = slab.material_table(sc.inv_mass[own], sc.params[own])
    ids = own.astype(np.uint32)
    static_boxes = sc.boxes
    if backend_name == "numpy":
        from oracle import oracle as O
        from tests.slab_backends import NumpySlabBackend
        O.build()
        be = NumpySlabBackend(O, sc.pos[own], sc.vel[own], index, ids, table, sc.bpos[bsel],
                              sc.bparams[bsel], sc.grid_size, static_boxes)
    else:
        from fluidsimulation_amd import native
        # "hip": every rank on device 0, records through torch.distributed (gloo, host staging);
        # "hip_native": one device per rank, records through the library's own RCCL communicator
        # (sph_slab_exchange); torch.distributed (gloo) then only carries the unique id and the verdict
        dev = rank if backend_name == "hip_native" else 0
        ctx = native.Context(device=dev)
        ctx.set_boundary(sc.bpos[bsel], sc.bparams[bsel])
        ctx.set_colliders(static_boxes)
        ctx.slab_upload(sc.pos[own], sc.vel[own], index, ids, table, sc.grid_size,
                        capacity=2 * sc.n)
        be = slab.HipSlabBackend(ctx, native.make_params(sc.grid_size), f"cuda:{dev}",
                                 buffer_records=sc.n, dynamic_bodies=bodies)
        if backend_name == "hip_native":
            be.init_native_exchange(rank, world)
    sim = slab.SlabSimulation(be, edges, halo=2.0 * h, rebalance_every=1 if rebalance else 0,
                              domain=(float(sc.pos[:, 0].min()), float(sc.pos[:, 0].max())),
                              max_shift=1.0 * h)
    history = []
    for _ in range(nsteps):
        sim.step()
        history.append(be.counts())
    be.synchronize()
    pos, vel, rho, pid = be.download()
    gathered = [None] * world
    my_boxes = be.boxes if backend_name == "numpy" else be.ctx.get_colliders()
    dist.gather_object((pos, vel, rho, pid, history, sim.sent_records, sim.transfers_out, my_boxes), gathered if rank == 0 else None)
    if rank == 0:
        allpos = np.concatenate([g[0] for g in gathered])
        allvel = np.concatenate([g[1] for g in gathered])
        allrho = np.concatenate([g[2] for g in gathered])
        allid = np.concatenate([g[3] for g in gathered]).astype(np.int64)
        verdict = {"world": world, "n": int(sc.n), "gathered": int(len(allid)),
                   "unique_ids": int(len(np.unique(allid))),
                   "sent_records": [int(g[5]) for g in gathered],
                   "owned_per_rank": [[h_["owned"] for h_ in g[4]] for g in gathered],
                   "ghosts_per_rank": [[h_["ghosts"] for h_ in g[4]] for g in gathered],
                   "final_edges": [e for e in sim.edges if np.isfinite(e)]}
        ok = verdict["gathered"] == sc.n and verdict["unique_ids"] == sc.n
        if ok:
            order = np.argsort(allid)
            gpos, gvel, grho = allpos[order], allvel[order], allrho[order]
            # the undecomposed run
            if backend_name == "numpy":
                from oracle import oracle as O
                rpos, rvel, rboxes = sc.pos.copy(), sc.vel.copy(), sc.boxes.copy()
                for _ in range(nsteps):
                    rvel, rrho, _, _ = O.sph_step(rpos, rvel, sc.inv_mass, sc.params, sc.bpos,
                                                  sc.bparams, sc.grid_size, dt=float(scene.DT))
                    rpos, rvel, rimp = O.integrate_particles(rpos, rvel, sc.inv_mass, sc.params,
                                                             rboxes, float(scene.DT))
                    rboxes = O.integrate_bodies(rboxes, rimp, float(scene.DT))
            else:
                from fluidsimulation_amd import native
                ref = native.Context(device=0)
                ref.set_boundary(sc.bpos, sc.bparams)
                ref.set_colliders(static_boxes)
                ref.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
                p = native.make_params(sc.grid_size)
                ref.step_resident(p, nsteps - 1)
                ref.hash_position(p); ref.radix_sort(); ref.map_table(); ref.reorder()
                ref.compute_density_and_pressure(p)
                rrho = ref.read(native.SPH_BUF_DENSITIES)
                ref.compute_force(p); ref.apply_force(p); ref.integrate(p)
                rpos, rvel = ref.download()
                rboxes = ref.get_colliders()
                ref.close()
            scale_p = float(np.abs(rpos).max())
            scale_v = float(np.abs(rvel).max())
            verdict["max_pos_err"] = float(np.abs(gpos - rpos).max() / scale_p)
            verdict["max_vel_err"] = float(np.abs(gvel - rvel).max() / scale_v)
            verdict["max_rho_err"] = float(np.abs(grho - rrho).max() / float(np.abs(rrho).max()))
            verdict["moved_owner"] = int(sum(g[6] for g in gathered))
            worst = np.argsort(-np.abs(gvel - rvel).max(axis=1))[:5]
            verdict["worst_particles"] = [[float(v) for v in rpos[i]] + [float(np.abs(gvel[i] - rvel[i]).max())] for i in worst]
            bscale = float(np.abs(rboxes).max())
            verdict["max_body_err"] = float(max(np.abs(g[7] - rboxes).max() for g in gathered) / bscale)
            verdict["body_moved"] = float(np.abs(rboxes - sc.boxes).max())
        verdict["ok"] = bool(ok)
        with open(out_path, "w") as f:
            json.dump(verdict, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
