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


def plan_cuts(edges, counts, halo, max_shift, domain=None):
    """New cut positions from the owned counts of all ranks (pure; every rank computes the same).
    Cut f moves by half of what would equalise the counts on its two sides, at most max_shift."""
    world = len(counts)
    total = sum(counts)
    if total == 0:
        return list(edges)
    target = total / world
    lo_dom, hi_dom = domain if domain is not None else (None, None)
    new = list(edges)
    for f in range(1, world):
        excess = sum(counts[:f]) - f * target        # > 0: too many particles left of cut f
        donor = f - 1 if excess > 0 else f
        lo, hi = edges[donor], edges[donor + 1]
        if donor == 0 and lo_dom is not None:
            lo = lo_dom
        if donor == world - 1 and hi_dom is not None:
            hi = hi_dom
        if not (math.isfinite(lo) and math.isfinite(hi)) or hi <= lo or counts[donor] == 0:
            continue
        per_unit = counts[donor] / (hi - lo)            # particles per unit of x in the donor
        shift = -0.5 * excess / per_unit
        new[f] = edges[f] + max(-max_shift, min(max_shift, shift))
    # A rank only talks to its two neighbours, so after the move everything within one halo of
    # a rank's slab must still be owned (before the exchange) by that rank or a neighbour:
    # the new cut f stays at least one halo inside the OLD slabs on both of its sides.
    for f in range(1, world):
        lo_ok = edges[f - 1] + halo if f > 1 else -math.inf
        hi_ok = edges[f + 1] - halo if f < world - 1 else math.inf
        new[f] = edges[f] if lo_ok > hi_ok else min(max(new[f], lo_ok), hi_ok)
    for f in range(2, world):                           # and the interior slabs >= one halo wide
        new[f] = max(new[f], new[f - 1] + halo)
    return new


class HipSlabBackend:
    """The product backend: libsphfluid (HIP) with torch tensors as exchange buffers."""

    def __init__(self, ctx, params, device, buffer_records, dynamic_bodies=False):
        self.ctx = ctx
        self.params = params
        self.dynamic_bodies = bool(dynamic_bodies)
        self.device = torch.device(device)
        self.cap = int(buffer_records)
        self._alloc(self.cap)
        self._keep = []        # received buffers the library may still be reading
        self.native_exchange = False

    def init_native_exchange(self, rank, world, group=None):
        """Opens the library's own RCCL communicator (sph_slab_comm_init): rank 0 draws the
        unique id, torch.distributed only carries its 128 bytes to the other ranks.  From then on
        SlabSimulation.exchange is one sph_slab_exchange call per step."""
        box = [native.Context.slab_comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                       group=group)
        self.ctx.slab_comm_init(rank, world, box[0])
        # the library's own send / receive buffers, sized like this backend's: no allocation
        # inside a step unless a message outgrows them
        self.ctx.slab_exchange_reserve(self.cap)
        self.native_exchange = True

    def _alloc(self, cap):
        self.cap = int(cap)
        self.left = torch.empty((self.cap, RECORD_FLOATS), dtype=torch.float32, device=self.device)
        self.right = torch.empty_like(self.left)

    def pack(self, xlo, xhi, halo):
        # sph_slab_pack synchronises the library's stream before it touches anything: every
        # kernel that read last step's received buffers has finished by the time it returns
        try:
            counts = self.ctx.slab_pack(xlo, xhi, halo, self.left.data_ptr(),
                                        self.right.data_ptr(), self.cap)
        except native.SphError as e:
            if e.code != native.SPH_E_TOOLARGE:
                raise
            # nothing was packed or re-flagged: size the buffers from the reported counts (the
            # torch stream is idle here: exchange() ends with a synchronise) and pack again
            need = self.ctx.last_slab_pack_counts
            torch.cuda.current_stream(self.device).synchronize()
            self._alloc(max(need[0], need[1]) * 3 // 2 + 64)
            counts = self.ctx.slab_pack(xlo, xhi, halo, self.left.data_ptr(),
                                        self.right.data_ptr(), self.cap)
        self._keep.clear()
        return self.left[:counts[0]], self.right[:counts[1]], counts

    def import_records(self, records, transfers):
        if records.shape[0] == 0:
            return
        assert records.is_contiguous() and records.device.type == self.device.type
        # the library reads the buffer on its own stream: the producer (torch) must be done
        torch.cuda.current_stream(self.device).synchronize()
        self.ctx.slab_import(records.data_ptr(), records.shape[0], transfers)
        self._keep.append(records)   # alive until the next pack has synchronised the stream

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

    def __init__(self, backend, edges, halo, rank=None, world=None, group=None,
                 rebalance_every=0, domain=None, max_shift=None):
        """rebalance_every = k > 0: every k steps the cuts move towards equal particle counts (by
        at most max_shift, default one halo); domain = (xmin, xmax) of the fluid, used for the
        width of the two open-ended slabs.  The static boundary particles of a rank must cover
        its slab plus 3 h (they enter the densities of its ghosts) wherever the cuts may go."""
        self.backend = backend
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        assert len(edges) == self.world + 1
        self.edges = [float(e) for e in edges]
        self.halo = float(halo)
        for r in range(1, self.world - 1):   # neighbour-only exchange: interior slabs >= one halo
            assert self.edges[r + 1] - self.edges[r] >= self.halo, "slab narrower than the halo"
        self._set_bounds()
        self.rebalance_every = int(rebalance_every)
        self.domain = domain
        self.max_shift = float(max_shift) if max_shift is not None else self.halo
        self.steps_done = 0
        self.owned_history = []
        # gloo moves host memory; nccl (RCCL) moves device memory
        self.stage_on_host = self.world > 1 and dist.get_backend(group) != "nccl"
        self.sent_records = 0
        self.received_records = 0
        self.transfers_out = 0

    def _set_bounds(self):
        self.xlo = self.edges[self.rank] if self.rank > 0 else -math.inf
        self.xhi = self.edges[self.rank + 1] if self.rank < self.world - 1 else math.inf

    def rebalance(self):
        """Moves the interior cuts towards equal owned counts.  Every rank computes the same new
        cuts from the all-gathered counts; the particles on the wrong side of a moved cut change
        owner through the ordinary transfer of the next exchange (the sender keeps ghost copies
        of what it hands over, both sides pack against the new cut, and plan_cuts keeps every
        rank's needs within reach of its two neighbours).  One all-gather of `world` integers, outside the per-step data path."""
        dev = torch.device("cpu") if self.stage_on_host else self.backend.device
        mine = torch.tensor([self.backend.counts()["owned"]], dtype=torch.int64, device=dev)
        gathered = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(gathered, mine, group=self.group)
        counts = [int(g.item()) for g in gathered]
        self.owned_history.append(counts)
        self.edges = plan_cuts(self.edges, counts, self.halo, self.max_shift, self.domain)
        self._set_bounds()

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
        elif left.is_cuda:
            # NCCL requests complete in stream order only: the next pack (on the library's own
            # stream) must not overwrite a send buffer that is still in flight
            torch.cuda.current_stream(left.device).synchronize()
        return result

    def exchange(self):
        if getattr(self.backend, "native_exchange", False):
            # pack, ncclSend / ncclRecv with both neighbours and import inside the library
            sl, sr, rl, rr = self.backend.ctx.slab_exchange(self.xlo, self.xhi, self.halo)
            self.sent_records += sl + sr
            self.received_records += rl + rr
            return
        left, right, counts = self.backend.pack(self.xlo, self.xhi, self.halo)
        self.transfers_out += counts[2] + counts[3]
        for records, transfers in self._exchange(left, right, counts):
            self.backend.import_records(records, transfers)

    def step(self, nsteps=1):
        for _ in range(nsteps):
            if self.world > 1 and self.rebalance_every > 0 and \
                    self.steps_done % self.rebalance_every == 0:
                self.rebalance()
            self.steps_done += 1
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
