"""Slab decomposition of the SPH step over several GPUs.

One process per GPU.  Rank r owns the fluid particles with x in [edges[r], edges[r+1]); once per
step it sends each x neighbour the particles that crossed the shared face (ownership transfer)
and a copy of those within `halo` = 2 h of it (ghosts), appends what it receives and runs the
ordinary resident step on owned + ghosts.  The ghosts are dropped by the next step's sort.

The exchange is two point-to-point messages per neighbour (a 2-word header, then the records)
through torch.distributed: RCCL send/recv of device buffers with the "nccl" backend, host staging
with "gloo".  There is no collective on the data path.

The per-rank solver sits behind a small backend interface so that the exchange logic can be
tested on CPU ranks (tests/ provides a numpy backend over the oracle); the product backend is
HipSlabBackend over libsphfluid's sph_slab_* entry points.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import native

RECORD_FLOATS = 8          # pos3, bits(material), vel3, bits(id | flags): 32 bytes
ID_GHOST = 0x80000000
ID_MASK = 0x3FFFFFFF


def material_table(inv_mass, params, group=None):
    """Deduplicates (radius, restDensity, viscosity, gasConstant, inverseMass) rows into a table
    that is the same on every rank (sorted union over the group) and returns (table, index)."""
    rows = np.concatenate([np.asarray(params, np.float32).reshape(-1, 4),
                           np.asarray(inv_mass, np.float32).reshape(-1, 1)], axis=1)
    local = np.unique(rows.view(np.uint32), axis=0) if len(rows) else np.zeros((0, 5), np.uint32)
    if group is not None or (dist.is_available() and dist.is_initialized()):
        gathered = [None] * dist.get_world_size(group)
        dist.all_gather_object(gathered, local, group=group)
        table = np.unique(np.concatenate(gathered, axis=0), axis=0)
    else:
        table = local
    if len(rows) == 0:
        return table.view(np.float32), np.zeros(0, np.uint32)
    # row -> index in the table (both are bit patterns, compared exactly)
    key = {tuple(r): i for i, r in enumerate(table.tolist())}
    bits = rows.view(np.uint32)
    change = np.ones(len(bits), bool)
    change[1:] = np.any(bits[1:] != bits[:-1], axis=1)
    index = np.empty(len(bits), np.uint32)
    starts = np.flatnonzero(change)
    ends = np.append(starts[1:], len(bits))
    for s, e in zip(starts, ends):   # runs of equal rows (materials come in large runs)
        index[s:e] = key[tuple(bits[s].tolist())]
    return np.ascontiguousarray(table.view(np.float32)), index


def uniform_edges(xmin, xmax, world):
    """world slabs of equal width; the outer two are open-ended."""
    e = [xmin + (xmax - xmin) * r / world for r in range(world + 1)]
    e[0], e[-1] = -math.inf, math.inf
    return e


class HipSlabBackend:
    """The product backend: libsphfluid (HIP) with torch tensors as exchange buffers."""

    def __init__(self, ctx, params, device, buffer_records, dynamic_bodies=False):
        self.ctx = ctx
        self.params = params
        self.dynamic_bodies = bool(dynamic_bodies)
        self.device = torch.device(device)
        self.cap = int(buffer_records)
        self.left = torch.empty((self.cap, RECORD_FLOATS), dtype=torch.float32, device=self.device)
        self.right = torch.empty_like(self.left)

    def pack(self, xlo, xhi, halo):
        counts = self.ctx.slab_pack(xlo, xhi, halo, self.left.data_ptr(), self.right.data_ptr(),
                                    self.cap)
        return self.left[:counts[0]], self.right[:counts[1]], counts

    def import_records(self, records, transfers):
        if records.shape[0] == 0:
            return
        assert records.is_contiguous() and records.device.type == self.device.type
        # the library reads the buffer on its own stream: the producer (torch) must be done
        torch.cuda.current_stream(self.device).synchronize()
        self.ctx.slab_import(records.data_ptr(), records.shape[0], transfers)
        self._keep = records   # stays alive until the next pack has synchronised the stream

    def step(self):
        self.ctx.step_resident(self.params, 1)

    # with dynamic boxes the step is split around the sum of the body impulses over ranks
    def step_particles(self):
        self.ctx.slab_step_particles(self.params)

    def get_impulses(self):
        return self.ctx.slab_get_impulses()

    def step_bodies(self, impulses):
        self.ctx.slab_step_bodies(self.params, impulses)

    def synchronize(self):
        self.ctx.synchronize()

    def download(self):
        return self.ctx.slab_download()

    def counts(self):
        return self.ctx.slab_counts()


class SlabSimulation:
    """Drives one rank of the decomposition.  backend: pack / import_records / step / download."""

    def __init__(self, backend, edges, halo, rank=None, world=None, group=None):
        self.backend = backend
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        assert len(edges) == self.world + 1
        self.xlo = float(edges[self.rank]) if self.rank > 0 else -math.inf
        self.xhi = float(edges[self.rank + 1]) if self.rank < self.world - 1 else math.inf
        self.halo = float(halo)
        # gloo moves host memory; nccl (RCCL) moves device memory
        self.stage_on_host = self.world > 1 and dist.get_backend(group) != "nccl"
        self.sent_records = 0
        self.received_records = 0
        self.transfers_out = 0

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _exchange(self, left, right, counts):
        """Sends `left` to rank-1 and `right` to rank+1; returns [(records, transfers), ...]."""
        peers = []
        if self.rank > 0:
            peers.append((self.rank - 1, left, counts[2]))
        if self.rank < self.world - 1:
            peers.append((self.rank + 1, right, counts[3]))
        if not peers:
            return []
        dev = torch.device("cpu") if self.stage_on_host else left.device
        # headers
        ops, hdr_in = [], []
        for peer, rec, transfers in peers:
            out = torch.tensor([rec.shape[0], transfers], dtype=torch.int64, device=dev)
            inn = torch.zeros(2, dtype=torch.int64, device=dev)
            hdr_in.append(inn)
            ops.append(dist.P2POp(dist.isend, out, self._peer(peer), self.group))
            ops.append(dist.P2POp(dist.irecv, inn, self._peer(peer), self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        # records
        ops, result, keep = [], [], []
        for (peer, rec, _), inn in zip(peers, hdr_in):
            n_in, t_in = (int(v) for v in inn.tolist())
            buf = torch.empty((n_in, RECORD_FLOATS), dtype=torch.float32, device=dev)
            result.append((buf, t_in))
            if rec.shape[0] > 0:
                out = rec.to(dev) if self.stage_on_host else rec
                keep.append(out)
                ops.append(dist.P2POp(dist.isend, out, self._peer(peer), self.group))
            if n_in > 0:
                ops.append(dist.P2POp(dist.irecv, buf, self._peer(peer), self.group))
            self.sent_records += rec.shape[0]
            self.received_records += n_in
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.stage_on_host:
            result = [(buf.to(left.device), t) for buf, t in result]
        return result

    def exchange(self):
        left, right, counts = self.backend.pack(self.xlo, self.xhi, self.halo)
        self.transfers_out += counts[2] + counts[3]
        for records, transfers in self._exchange(left, right, counts):
            self.backend.import_records(records, transfers)

    def step(self, nsteps=1):
        for _ in range(nsteps):
            if self.world > 1:
                self.exchange()
            if getattr(self.backend, "dynamic_bodies", False):
                # replicated rigid bodies: every rank needs the pushes of all ranks' particles;
                # the one collective of the step, K*6 floats
                self.backend.step_particles()
                imp = torch.from_numpy(np.ascontiguousarray(self.backend.get_impulses()))
                if self.world > 1:
                    if not self.stage_on_host:
                        imp = imp.to(self.backend.device)
                    dist.all_reduce(imp, op=dist.ReduceOp.SUM, group=self.group)
                self.backend.step_bodies(imp.cpu().numpy())
            else:
                self.backend.step()


def split_scene(scene, edges, rank, margin):
    """The part of a whole scene that rank `rank` starts with: its fluid particles (with their
    global index as id) and the boundary particles within `margin` of its slab."""
    world = len(edges) - 1
    lo = edges[rank] if rank > 0 else -math.inf
    hi = edges[rank + 1] if rank < world - 1 else math.inf
    x = scene.pos[:, 0]
    own = np.flatnonzero((x >= lo) & (x < hi))
    bx = scene.bpos[:, 0]
    bsel = np.flatnonzero((bx >= lo - margin) & (bx < hi + margin))
    return own, bsel
