"""CPU stand-in for the per-rank solver of fluidsimulation_amd.slab, for the gloo tests.

TEST INFRASTRUCTURE: it steps the particles with the oracle (oracle/sph_oracle.c) and restates
the pack / import semantics of sph_slab_pack / sph_slab_import (include/sphfluid.h) in numpy, so
that SlabSimulation's exchange logic runs on ranks that have no GPU."""
import numpy as np
import torch

from fluidsimulation_amd import scene
from fluidsimulation_amd.slab import ID_GHOST, ID_MASK, RECORD_FLOATS


class NumpySlabBackend:
    def __init__(self, oracle, pos, vel, mat_index, ids, materials, bpos, bparams, grid_size,
                 boxes, dt=float(scene.DT)):
        self.o = oracle
        self.pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3).copy()
        self.vel = np.ascontiguousarray(vel, np.float32).reshape(-1, 3).copy()
        self.mat = np.ascontiguousarray(mat_index, np.uint32).copy()
        self.ids = np.ascontiguousarray(ids, np.uint32).copy()       # with the ghost bit
        self.materials = np.ascontiguousarray(materials, np.float32).reshape(-1, 5)
        self.bpos, self.bparams = bpos, bparams
        self.grid, self.boxes, self.dt = grid_size, boxes, dt
        self.rho = np.zeros(len(self.pos), np.float32)
        self.boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, scene.BOX_FLOATS).copy()
        self.dynamic_bodies = bool(np.any(self.boxes[:, 16:20] != 0))
        self.transfers_out = 0
        self.transfers_in = 0

    def _records(self, sel, ghost):
        rec = np.empty((len(sel), RECORD_FLOATS), np.float32)
        rec[:, 0:3] = self.pos[sel]
        rec[:, 3] = self.mat[sel].view(np.float32)
        rec[:, 4:7] = self.vel[sel]
        ids = self.ids[sel] | np.where(ghost, np.uint32(ID_GHOST), np.uint32(0)).astype(np.uint32)
        rec[:, 7] = ids.view(np.float32)
        return torch.from_numpy(rec)

    def pack(self, xlo, xhi, halo):
        keep = (self.ids & ID_GHOST) == 0          # last step's ghosts die
        self.pos, self.vel, self.mat, self.ids = (a[keep] for a in
                                                  (self.pos, self.vel, self.mat, self.ids))
        x = self.pos[:, 0]
        move_l, move_r = x < np.float32(xlo), x >= np.float32(xhi)
        send_l = x < np.float32(xlo) + np.float32(halo) if np.isfinite(xlo) else move_l
        send_r = x >= np.float32(xhi) - np.float32(halo) if np.isfinite(xhi) else move_r
        sl, sr = np.flatnonzero(send_l), np.flatnonzero(send_r)
        left = self._records(sl, ~move_l[sl])
        right = self._records(sr, ~move_r[sr])
        moved = move_l | move_r
        self.ids = np.where(moved, self.ids | np.uint32(ID_GHOST), self.ids).astype(np.uint32)
        counts = [len(sl), len(sr), int(move_l.sum()), int(move_r.sum())]
        self.transfers_out += counts[2] + counts[3]
        return left, right, counts

    def import_records(self, records, transfers):
        rec = records.numpy()
        if len(rec) == 0:
            return
        ids = np.ascontiguousarray(rec[:, 7]).view(np.uint32)
        assert int(((ids & ID_GHOST) == 0).sum()) == transfers
        self.transfers_in += transfers
        self.pos = np.concatenate([self.pos, rec[:, 0:3]])
        self.vel = np.concatenate([self.vel, rec[:, 4:7]])
        self.mat = np.concatenate([self.mat, np.ascontiguousarray(rec[:, 3]).view(np.uint32)])
        self.ids = np.concatenate([self.ids, ids])

    def step_particles(self):
        m = self.materials[self.mat]
        params, inv_mass = np.ascontiguousarray(m[:, 0:4]), np.ascontiguousarray(m[:, 4])
        vel, dens, _, _ = self.o.sph_step(self.pos, self.vel, inv_mass, params, self.bpos,
                                          self.bparams, self.grid, dt=self.dt)
        own = (self.ids & ID_GHOST) == 0   # ghosts do not push the bodies
        _, _, self.impulses = self.o.integrate_particles(self.pos[own], vel[own], inv_mass[own],
                                                         params[own], self.boxes, self.dt)
        self.pos, self.vel, _ = self.o.integrate_particles(self.pos, vel, inv_mass, params,
                                                           self.boxes, self.dt)
        self.rho = dens

    def get_impulses(self):
        return self.impulses

    def step_bodies(self, impulses):
        self.boxes = self.o.integrate_bodies(self.boxes, impulses, self.dt)

    def step(self):
        self.step_particles()
        self.step_bodies(self.impulses)

    def synchronize(self):
        pass

    def download(self):
        own = (self.ids & ID_GHOST) == 0
        return self.pos[own], self.vel[own], self.rho[own], self.ids[own] & ID_MASK

    def counts(self):
        own = int(((self.ids & ID_GHOST) == 0).sum())
        return {"resident": len(self.ids), "owned": own, "ghosts": len(self.ids) - own, "dead": 0}
