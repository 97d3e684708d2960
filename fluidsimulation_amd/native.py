"""ctypes binding of libsphfluid.so (include/sphfluid.h), 1:1 with the C ABI.

This is what a P/Invoke layer in the reference's C# host would look like; here it serves the
Python harnesses (tests, bench).  There is no fallback: if the shared library is missing, import
fails loudly (build it with `python -m fluidsimulation_amd.build` or __graft_entry__.build()).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPHFLUID_LIB selects an experimental build of the same ABI (kernel tuning only)
LIB_PATH = os.environ.get("SPHFLUID_LIB") or os.path.join(_HERE, "libsphfluid.so")

SPH_OK = 0
SPH_E_INVALID, SPH_E_NOMEM, SPH_E_DEVICE, SPH_E_NODEVICE, SPH_E_TOOLARGE = -1, -2, -3, -4, -5

(SPH_T_KEYS, SPH_T_SORT, SPH_T_TABLE, SPH_T_REORDER, SPH_T_DENSITY, SPH_T_FORCE, SPH_T_INTEGRATE,
 SPH_T_BODIES, SPH_T_PLAN, SPH_T_COUNT) = range(10)
STAGE_NAMES = ["keys", "sort", "table", "reorder", "density", "force", "integrate", "bodies", "plan"]

(SPH_BUF_HASHES, SPH_BUF_CELLS, SPH_BUF_KEYS, SPH_BUF_PERM, SPH_BUF_SORTED_KEYS, SPH_BUF_TABLE,
 SPH_BUF_DENSITIES, SPH_BUF_PRESSURES, SPH_BUF_FORCES, SPH_BUF_DENSITY_COUNTS,
 SPH_BUF_FORCE_COUNTS, SPH_BUF_POSITIONS, SPH_BUF_VELOCITIES, SPH_BUF_BOUNDARY_PERM,
 SPH_BUF_BOUNDARY_TABLE, SPH_BUF_MASK_COUNTS, SPH_BUF_MASK_IDSUMS) = range(17)

SPH_DEBUG_HASHES, SPH_DEBUG_COUNTS, SPH_DEBUG_NO_LDS_TILE = 1, 2, 4
SPH_SORT_NOALLOC_MAX = 1 << 20

# every symbol include/sphfluid.h declares (tests check that the library exports them all)
EXPORTS = [
    "sph_create", "sph_destroy", "sph_last_error", "sph_device_name", "sph_set_debug",
    "sph_synchronize", "sph_find_grid_size", "sph_set_boundary", "sph_set_colliders",
    "sph_get_colliders", "sph_set_grid", "sph_get_grid", "sph_upload", "sph_download",
    "sph_seed_volume", "sph_unpin_host", "sph_pin_stats",
    "sph_hash_position", "sph_radix_sort", "sph_map_table", "sph_reorder",
    "sph_compute_density_and_pressure", "sph_compute_force", "sph_apply_force", "sph_integrate",
    "sph_step_resident", "sph_step", "sph_step_compat", "sph_read", "sph_find_bounds",
    "sph_sample_density", "sph_sort_pairs", "sph_sort_pairs_noalloc", "sph_get_timings", "sph_reset_timings",
    "sph_slab_upload", "sph_slab_pack", "sph_slab_import", "sph_slab_counts", "sph_slab_download",
    "sph_slab_step_particles", "sph_slab_get_impulses", "sph_slab_step_bodies",
    "sph_slab_comm_unique_id", "sph_slab_comm_init", "sph_slab_exchange", "sph_slab_comm_destroy",
    "sph_slab_exchange_reserve",
]


class SphParams(C.Structure):
    _fields_ = [("dt", C.c_float), ("kernelRadiusRate", C.c_float), ("gravity", C.c_float * 3),
                ("gridSize", C.c_float), ("particleRestitution", C.c_float),
                ("boundBodies", C.c_int32), ("bodyMin", C.c_float * 3),
                ("bodyMax", C.c_float * 3), ("bodyContacts", C.c_int32)]


class SphConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("maxCells", C.c_int32), ("enableTimings", C.c_int32),
                ("pinHostArrays", C.c_int32)]


class SphGrid(C.Structure):
    _fields_ = [("origin", C.c_int32 * 3), ("dims", C.c_int32 * 3), ("keyBits", C.c_int32),
                ("cells", C.c_int32)]


class SphTimings(C.Structure):
    _fields_ = [("ms", C.c_double * SPH_T_COUNT), ("launches", C.c_int64 * SPH_T_COUNT),
                ("steps", C.c_int64)]


class SphError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libsphfluid error {code}: {text}")
        self.code = code


_lib = None


def load():
    """dlopen libsphfluid.so and declare the prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP/HSA runtime.  A process that uses both must load torch's copy
    # first: loaded after libsphfluid's, it finds no device ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run __graft_entry__.build() "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.sph_create.argtypes = [C.POINTER(SphConfig), C.POINTER(vp)]
    lib.sph_destroy.argtypes = [vp]
    lib.sph_last_error.argtypes = [vp]
    lib.sph_last_error.restype = C.c_char_p
    lib.sph_device_name.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32)]
    lib.sph_set_debug.argtypes = [vp, i32]
    lib.sph_synchronize.argtypes = [vp]
    lib.sph_find_grid_size.argtypes = [vp, vp, i32, f32]
    lib.sph_find_grid_size.restype = f32
    lib.sph_set_boundary.argtypes = [vp, vp, vp, i32]
    lib.sph_set_colliders.argtypes = [vp, vp, i32]
    lib.sph_get_colliders.argtypes = [vp, vp, i32]
    lib.sph_set_grid.argtypes = [vp, C.POINTER(SphGrid)]
    lib.sph_get_grid.argtypes = [vp, C.POINTER(SphGrid)]
    lib.sph_upload.argtypes = [vp, vp, vp, vp, vp, i32, f32]
    lib.sph_download.argtypes = [vp, vp, vp]
    lib.sph_unpin_host.argtypes = [vp, vp]
    lib.sph_pin_stats.argtypes = [vp, C.POINTER(i32 * 3)]
    lib.sph_seed_volume.argtypes = [vp, vp, vp, vp, f32, vp, f32, f32, i32, C.POINTER(i32)]
    for name in ("sph_hash_position", "sph_compute_density_and_pressure", "sph_compute_force",
                 "sph_apply_force", "sph_integrate"):
        getattr(lib, name).argtypes = [vp, C.POINTER(SphParams)]
    for name in ("sph_radix_sort", "sph_map_table", "sph_reorder", "sph_reset_timings"):
        getattr(lib, name).argtypes = [vp]
    lib.sph_step_resident.argtypes = [vp, C.POINTER(SphParams), i32]
    lib.sph_step.argtypes = [vp, C.POINTER(SphParams), vp, vp, vp, vp, i32, vp, vp]
    lib.sph_step_compat.argtypes = [vp, C.POINTER(SphParams), vp, vp, vp, vp, i32, vp, vp]
    lib.sph_read.argtypes = [vp, i32, vp, C.c_size_t]
    lib.sph_find_bounds.argtypes = [vp, f32, vp, vp]
    lib.sph_sample_density.argtypes = [vp, C.POINTER(SphParams), vp, f32, vp, vp]
    lib.sph_sort_pairs.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.sph_sort_pairs_noalloc.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.sph_get_timings.argtypes = [vp, C.POINTER(SphTimings)]
    lib.sph_slab_upload.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, f32, i32]
    lib.sph_slab_pack.argtypes = [vp, f32, f32, f32, vp, vp, i32, C.POINTER(i32 * 4)]
    lib.sph_slab_import.argtypes = [vp, vp, i32, i32]
    lib.sph_slab_counts.argtypes = [vp, C.POINTER(i32 * 4)]
    lib.sph_slab_step_particles.argtypes = [vp, C.POINTER(SphParams)]
    lib.sph_slab_get_impulses.argtypes = [vp, vp, i32]
    lib.sph_slab_step_bodies.argtypes = [vp, C.POINTER(SphParams), vp, i32]
    lib.sph_slab_download.argtypes = [vp, vp, vp, vp, vp, i32, C.POINTER(i32)]
    lib.sph_slab_comm_unique_id.argtypes = [vp, i32]
    lib.sph_slab_comm_init.argtypes = [vp, i32, i32, vp, i32]
    lib.sph_slab_exchange.argtypes = [vp, f32, f32, f32, C.POINTER(i32 * 4)]
    lib.sph_slab_comm_destroy.argtypes = [vp]
    lib.sph_slab_exchange_reserve.argtypes = [vp, i32]
    for name in EXPORTS:
        if name not in ("sph_last_error", "sph_find_grid_size"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def make_params(grid_size, dt=1.0 / 60.0, rate=4.0, gravity=(0.0, -9.81, 0.0),
                particle_restitution=0.5, body_bounds=None, body_contacts=False):
    p = SphParams()
    p.dt = dt
    p.kernelRadiusRate = rate
    p.gravity[:] = gravity
    p.gridSize = grid_size
    p.particleRestitution = particle_restitution
    p.bodyContacts = 1 if body_contacts else 0
    if body_bounds is not None:
        p.boundBodies = 1
        p.bodyMin[:] = [float(v) for v in body_bounds[0]]
        p.bodyMax[:] = [float(v) for v in body_bounds[1]]
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


class Context:
    """Thin RAII wrapper over SphCtx*; methods map 1:1 to the C entry points."""

    def __init__(self, device=0, max_cells=0, timings=False, pin_host_arrays=False):
        """pin_host_arrays: SphConfig.pinHostArrays -- only with arrays that come back every frame
        and outlive the context (step_inplace), never with temporaries"""
        self.lib = load()
        cfg = SphConfig(device, max_cells, int(timings), 1 if pin_host_arrays else 0)
        h = C.c_void_p()
        rc = self.lib.sph_create(C.byref(cfg), C.byref(h))
        if rc != SPH_OK:
            raise SphError(rc, self.lib.sph_last_error(None).decode())
        self.h = h
        self.n = 0
        self.nboxes = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.sph_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != SPH_OK:
            raise SphError(rc, self.lib.sph_last_error(self.h).decode())

    # -- scene state
    def device_name(self):
        buf = C.create_string_buffer(256)
        cus = C.c_int32()
        self._chk(self.lib.sph_device_name(self.h, buf, 256, C.byref(cus)))
        return buf.value.decode(), cus.value

    def set_debug(self, flags):
        self._chk(self.lib.sph_set_debug(self.h, flags))

    def synchronize(self):
        self._chk(self.lib.sph_synchronize(self.h))

    def set_boundary(self, bpos, bparams):
        bpos = _f32(bpos, (-1, 3))
        bparams = _f32(bparams, (-1, 3))
        self._chk(self.lib.sph_set_boundary(self.h, _ptr(bpos), _ptr(bparams), len(bpos)))

    def set_colliders(self, boxes):
        boxes = _f32(boxes, (-1, 25))
        self.nboxes = len(boxes)
        self._chk(self.lib.sph_set_colliders(self.h, _ptr(boxes), len(boxes)))

    def get_colliders(self):
        out = np.empty((self.nboxes, 25), np.float32)
        self._chk(self.lib.sph_get_colliders(self.h, _ptr(out), self.nboxes))
        return out

    def set_grid(self, origin, dims):
        if origin is None:
            self._chk(self.lib.sph_set_grid(self.h, None))
            return
        g = SphGrid()
        g.origin[:] = [int(v) for v in origin]
        g.dims[:] = [int(v) for v in dims]
        self._chk(self.lib.sph_set_grid(self.h, C.byref(g)))

    def get_grid(self):
        g = SphGrid()
        self._chk(self.lib.sph_get_grid(self.h, C.byref(g)))
        return list(g.origin), list(g.dims), g.keyBits, g.cells

    def upload(self, pos, vel, inv_mass, params, grid_size):
        pos, vel = _f32(pos, (-1, 3)), _f32(vel, (-1, 3))
        inv_mass, params = _f32(inv_mass, (-1,)), _f32(params, (-1, 4))
        self.n = len(pos)
        assert len(vel) == self.n and len(inv_mass) == self.n and len(params) == self.n
        self._chk(self.lib.sph_upload(self.h, _ptr(pos), _ptr(vel), _ptr(inv_mass), _ptr(params),
                                      self.n, grid_size))

    def seed_volume(self, local_to_world, bounds_min, bounds_size, radius, params, inv_mass,
                    grid_size, append=False):
        l2w = _f32(local_to_world, (12,))
        bmin, bsize = _f32(bounds_min, (3,)), _f32(bounds_size, (3,))
        prm = _f32(params, (4,))
        added = C.c_int32()
        self._chk(self.lib.sph_seed_volume(self.h, _ptr(l2w), _ptr(bmin), _ptr(bsize), radius,
                                           _ptr(prm), inv_mass, grid_size, 1 if append else 0,
                                           C.byref(added)))
        self.n = (self.n if append else 0) + added.value
        return added.value

    def download(self):
        pos = np.empty((self.n, 3), np.float32)
        vel = np.empty((self.n, 3), np.float32)
        self._chk(self.lib.sph_download(self.h, _ptr(pos), _ptr(vel)))
        return pos, vel

    def unpin_host(self, array=None):
        """release the registration of one caller array (None: of all) before it is freed"""
        self._chk(self.lib.sph_unpin_host(self.h, _ptr(array)))

    def pin_stats(self):
        st = (C.c_int32 * 3)()
        self._chk(self.lib.sph_pin_stats(self.h, C.byref(st)))
        return dict(zip(("registered", "succeeded", "failed"), st))

    # -- stages (FluidSPHByGPUSystem.cs:242-247 names)
    def hash_position(self, p):
        self._chk(self.lib.sph_hash_position(self.h, C.byref(p)))

    def radix_sort(self):
        self._chk(self.lib.sph_radix_sort(self.h))

    def map_table(self):
        self._chk(self.lib.sph_map_table(self.h))

    def reorder(self):
        self._chk(self.lib.sph_reorder(self.h))

    def compute_density_and_pressure(self, p):
        self._chk(self.lib.sph_compute_density_and_pressure(self.h, C.byref(p)))

    def compute_force(self, p):
        self._chk(self.lib.sph_compute_force(self.h, C.byref(p)))

    def apply_force(self, p):
        self._chk(self.lib.sph_apply_force(self.h, C.byref(p)))

    def integrate(self, p):
        self._chk(self.lib.sph_integrate(self.h, C.byref(p)))

    def step_resident(self, p, nsteps=1):
        self._chk(self.lib.sph_step_resident(self.h, C.byref(p), nsteps))

    def step(self, p, pos, vel, inv_mass, params, want_densities=False, want_forces=False):
        pos, vel = _f32(pos, (-1, 3)).copy(), _f32(vel, (-1, 3)).copy()
        inv_mass, params = _f32(inv_mass, (-1,)), _f32(params, (-1, 4))
        n = len(pos)
        self.n = n
        dens = np.empty(n, np.float32) if want_densities else None
        forces = np.empty((n, 3), np.float32) if want_forces else None
        self._chk(self.lib.sph_step(self.h, C.byref(p), _ptr(pos), _ptr(vel), _ptr(inv_mass),
                                    _ptr(params), n, _ptr(dens), _ptr(forces)))
        return pos, vel, dens, forces

    def sample_density(self, p, origin, spacing, dims):
        """density field rho(x) on a regular grid (z, y, x order of the returned array)"""
        o = np.ascontiguousarray(origin, np.float32)
        d = np.ascontiguousarray(dims, np.int32)
        out = np.empty((int(d[2]), int(d[1]), int(d[0])), np.float32)
        self._chk(self.lib.sph_sample_density(self.h, C.byref(p), _ptr(o), float(spacing), _ptr(d), _ptr(out)))
        return out

    def step_inplace(self, p, pos, vel, inv_mass, params, dens=None, forces=None):
        """sph_step on the caller's own arrays, as the C# host calls it: pos / vel are read and
        overwritten, nothing is copied on the Python side"""
        for a in (pos, vel, inv_mass, params):
            assert a.dtype == np.float32 and a.flags.c_contiguous
        n = len(pos)
        self.n = n
        self._chk(self.lib.sph_step(self.h, C.byref(p), _ptr(pos), _ptr(vel), _ptr(inv_mass),
                                    _ptr(params), n, _ptr(dens), _ptr(forces)))

    def step_compat(self, p, pos, vel, inv_mass, params):
        pos, vel = _f32(pos, (-1, 3)), _f32(vel, (-1, 3))
        inv_mass, params = _f32(inv_mass, (-1,)), _f32(params, (-1, 4))
        n = len(pos)
        self.n = n
        dens = np.empty(n, np.float32)
        forces = np.empty((n, 3), np.float32)
        self._chk(self.lib.sph_step_compat(self.h, C.byref(p), _ptr(pos), _ptr(vel),
                                           _ptr(inv_mass), _ptr(params), n, _ptr(dens),
                                           _ptr(forces)))
        return dens, forces

    # -- inspection
    def read(self, buffer):
        n = self.n
        if buffer in (SPH_BUF_HASHES, SPH_BUF_KEYS, SPH_BUF_PERM, SPH_BUF_SORTED_KEYS):
            out = np.empty(n, np.uint32)
        elif buffer == SPH_BUF_CELLS:
            out = np.empty((n, 3), np.int32)
        elif buffer in (SPH_BUF_TABLE, SPH_BUF_BOUNDARY_TABLE):
            out = np.empty((self.get_grid()[3], 2), np.int32)
        elif buffer in (SPH_BUF_DENSITIES, SPH_BUF_PRESSURES):
            out = np.empty(n, np.float32)
        elif buffer in (SPH_BUF_DENSITY_COUNTS, SPH_BUF_FORCE_COUNTS, SPH_BUF_MASK_COUNTS):
            out = np.empty(n, np.int32)
        elif buffer == SPH_BUF_MASK_IDSUMS:
            out = np.empty(n, np.uint64)
        elif buffer in (SPH_BUF_FORCES, SPH_BUF_POSITIONS, SPH_BUF_VELOCITIES):
            out = np.empty((n, 3), np.float32)
        else:
            raise ValueError(buffer)
        self._chk(self.lib.sph_read(self.h, buffer, _ptr(out), out.nbytes))
        return out

    def read_boundary_perm(self, b):
        out = np.empty(b, np.uint32)
        self._chk(self.lib.sph_read(self.h, SPH_BUF_BOUNDARY_PERM, _ptr(out), out.nbytes))
        return out

    def find_bounds(self, grid_size):
        mn = np.empty(3, np.float32)
        mx = np.empty(3, np.float32)
        self._chk(self.lib.sph_find_bounds(self.h, grid_size, _ptr(mn), _ptr(mx)))
        return mn, mx

    def sort_pairs(self, keys, key_bits=32):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        perm = np.empty(len(keys), np.uint32)
        skeys = np.empty(len(keys), np.uint32)
        self._chk(self.lib.sph_sort_pairs(self.h, _ptr(keys), len(keys), key_bits, _ptr(perm),
                                          _ptr(skeys)))
        return perm, skeys

    def sort_pairs_noalloc(self, keys_ptr, n, key_bits, perm_out_ptr, keys_out_ptr):
        """device pointers (plain integers); asynchronous on the context's stream"""
        self._chk(self.lib.sph_sort_pairs_noalloc(self.h, keys_ptr, n, key_bits, perm_out_ptr,
                                                  keys_out_ptr))

    def timings(self):
        t = SphTimings()
        self._chk(self.lib.sph_get_timings(self.h, C.byref(t)))
        return {"ms": {STAGE_NAMES[i]: t.ms[i] for i in range(SPH_T_COUNT)},
                "launches": {STAGE_NAMES[i]: t.launches[i] for i in range(SPH_T_COUNT)},
                "steps": t.steps}

    def reset_timings(self):
        self._chk(self.lib.sph_reset_timings(self.h))

    # -- slab decomposition (device pointers are plain integers, e.g. torch.Tensor.data_ptr())
    def slab_upload(self, pos, vel, material_index, ids, materials, grid_size, capacity):
        pos, vel = _f32(pos, (-1, 3)), _f32(vel, (-1, 3))
        mi = np.ascontiguousarray(material_index, dtype=np.uint32)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        mats = _f32(materials, (-1, 5))
        self.n = len(pos)
        assert len(vel) == self.n and len(mi) == self.n and len(ids) == self.n
        self._chk(self.lib.sph_slab_upload(self.h, _ptr(pos), _ptr(vel), _ptr(mi), _ptr(ids),
                                           self.n, _ptr(mats), len(mats), grid_size, capacity))

    def slab_pack(self, xlo, xhi, halo, left_ptr, right_ptr, cap_records):
        counts = (C.c_int32 * 4)()
        rc = self.lib.sph_slab_pack(self.h, xlo, xhi, halo, left_ptr, right_ptr, cap_records,
                                    C.byref(counts))
        self.last_slab_pack_counts = list(counts)   # on SPH_E_TOOLARGE: what the buffers need
        self._chk(rc)
        return list(counts)

    @staticmethod
    def slab_comm_unique_id():
        """128 bytes from ncclGetUniqueId (rank 0; the launcher carries them to every rank)"""
        buf = (C.c_uint8 * 128)()
        n = load().sph_slab_comm_unique_id(buf, 128)
        if n < 0:
            raise SphError(n, load().sph_last_error(None).decode())
        return bytes(buf[:n])

    def slab_comm_init(self, rank, world, unique_id):
        buf = (C.c_uint8 * len(unique_id)).from_buffer_copy(unique_id)
        self._chk(self.lib.sph_slab_comm_init(self.h, rank, world, buf, len(unique_id)))

    def slab_exchange(self, xlo, xhi, halo):
        """pack + RCCL send/recv with both x neighbours + import, inside the library"""
        counts = (C.c_int32 * 4)()
        self._chk(self.lib.sph_slab_exchange(self.h, xlo, xhi, halo, C.byref(counts)))
        return list(counts)

    def slab_exchange_reserve(self, records):
        self._chk(self.lib.sph_slab_exchange_reserve(self.h, int(records)))

    def slab_comm_destroy(self):
        self._chk(self.lib.sph_slab_comm_destroy(self.h))

    def slab_import(self, records_ptr, count, transfers):
        self._chk(self.lib.sph_slab_import(self.h, records_ptr, count, transfers))

    def slab_step_particles(self, p):
        self._chk(self.lib.sph_slab_step_particles(self.h, C.byref(p)))

    def slab_get_impulses(self):
        out = np.zeros((self.nboxes, 6), np.float32)
        self._chk(self.lib.sph_slab_get_impulses(self.h, _ptr(out), self.nboxes))
        return out

    def slab_step_bodies(self, p, impulses=None):
        imp = None if impulses is None else _f32(impulses, (-1, 6))
        self._chk(self.lib.sph_slab_step_bodies(self.h, C.byref(p), _ptr(imp), self.nboxes))

    def slab_counts(self):
        counts = (C.c_int32 * 4)()
        self._chk(self.lib.sph_slab_counts(self.h, C.byref(counts)))
        return dict(zip(("resident", "owned", "ghosts", "dead"), counts))

    def slab_download(self):
        cap = self.slab_counts()["resident"]
        pos = np.empty((cap, 3), np.float32)
        vel = np.empty((cap, 3), np.float32)
        rho = np.empty(cap, np.float32)
        ids = np.empty(cap, np.uint32)
        k = C.c_int32()
        self._chk(self.lib.sph_slab_download(self.h, _ptr(pos), _ptr(vel), _ptr(rho), _ptr(ids),
                                             cap, C.byref(k)))
        k = k.value
        return pos[:k], vel[:k], rho[:k], ids[:k]


def find_grid_size(params, chunk_first, rate=4.0):
    lib = load()
    params = _f32(params, (-1, 4))
    chunk_first = np.ascontiguousarray(chunk_first, dtype=np.int32)
    return float(lib.sph_find_grid_size(_ptr(params), _ptr(chunk_first), len(chunk_first), rate))
