"""ctypes front end of the CPU oracle (oracle/sph_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under fluidsimulation_amd/ may import this module.
PARITY UNPINNED -- see the header of sph_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BOX_FLOATS = 25  # OrcBox == SphBox (include/sphfluid.h)


def build(force=False):
    so = os.path.join(_HERE, "libsphoracle.so")
    src = os.path.join(_HERE, "sph_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsphoracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsphoracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_hash.restype = C.c_uint32
        _LIB.orc_hash.argtypes = [C.c_int32] * 3
        _LIB.orc_fnv1_bytes.restype = C.c_uint32
        _LIB.orc_find_grid_size.restype = C.c_float
        _LIB.orc_count_near_hash_collisions.restype = C.c_long
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fnv1_bytes(data: bytes) -> int:
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    return int(lib().orc_fnv1_bytes(buf, C.c_int(len(data))))


def hash_cell(x, y, z) -> int:
    return int(lib().orc_hash(int(x), int(y), int(z)))


def quantize(pos, size):
    pos = _f32(pos).reshape(-1, 3)
    cells = np.empty(pos.shape, dtype=np.int32)
    lib().orc_quantize(_p(pos), C.c_int(len(pos)), C.c_float(size), _p(cells))
    return cells


def hash_cells(cells):
    cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
    out = np.empty(len(cells), dtype=np.uint32)
    lib().orc_hash_cells(_p(cells), C.c_int(len(cells)), _p(out))
    return out


def find_grid_size(chunk_first_radius, rate=4.0):
    r = _f32(chunk_first_radius).reshape(-1)
    return float(lib().orc_find_grid_size(_p(r), C.c_int(len(r)), C.c_float(rate)))


def find_bounds(pos, bpos, grid):
    pos = _f32(pos).reshape(-1, 3)
    bpos = _f32(bpos).reshape(-1, 3)
    mn = np.empty(3, np.float32)
    mx = np.empty(3, np.float32)
    lib().orc_find_bounds(_p(pos), C.c_int(len(pos)), _p(bpos), C.c_int(len(bpos)),
                          C.c_float(grid), _p(mn), _p(mx))
    return mn, mx


def density_pressure(pos, inv_mass, params, bpos, grid, rate=4.0, count=True, nthreads=0):
    pos = _f32(pos).reshape(-1, 3)
    bpos = _f32(bpos).reshape(-1, 3)
    inv_mass = _f32(inv_mass)
    params = _f32(params).reshape(-1, 4)
    n = len(pos)
    dens = np.empty(n, np.float32)
    pres = np.empty(n, np.float32)
    cnt = np.zeros(n, np.int32)
    rc = lib().orc_density_pressure(_p(pos), _p(inv_mass), _p(params), C.c_int(n), _p(bpos),
                                    C.c_int(len(bpos)), C.c_float(grid), C.c_float(rate),
                                    _p(dens), _p(pres), _p(cnt) if count else None,
                                    C.c_int(nthreads))
    assert rc == 0
    return dens, pres, cnt


def force(pos, inv_mass, vel, params, bpos, bparams, dens, pres, grid, rate=4.0, count=True,
          nthreads=0):
    pos = _f32(pos).reshape(-1, 3)
    vel = _f32(vel).reshape(-1, 3)
    bpos = _f32(bpos).reshape(-1, 3)
    bparams = _f32(bparams).reshape(-1, 3)
    inv_mass = _f32(inv_mass)
    params = _f32(params).reshape(-1, 4)
    dens = _f32(dens)
    pres = _f32(pres)
    n = len(pos)
    out = np.empty((n, 3), np.float32)
    cnt = np.zeros(n, np.int32)
    rc = lib().orc_force(_p(pos), _p(inv_mass), _p(vel), _p(params), C.c_int(n), _p(bpos),
                         _p(bparams), C.c_int(len(bpos)), _p(dens), _p(pres), C.c_float(grid),
                         C.c_float(rate), _p(out), _p(cnt) if count else None, C.c_int(nthreads))
    assert rc == 0
    return out, cnt


def apply_force(vel, forces, dens, inv_mass, dt):
    vel = _f32(vel).reshape(-1, 3).copy()
    forces = _f32(forces).reshape(-1, 3)
    dens = _f32(dens)
    inv_mass = _f32(inv_mass)
    lib().orc_apply_force(_p(vel), _p(forces), _p(dens), _p(inv_mass), C.c_int(len(vel)),
                          C.c_float(dt))
    return vel


def sph_step(pos, vel, inv_mass, params, bpos, bparams, grid, rate=4.0, dt=1.0 / 60.0,
             nthreads=0):
    """One FluidSPHByCPUSystem.OnUpdate: returns (new_vel, dens, pres, forces)."""
    pos = _f32(pos).reshape(-1, 3)
    vel = _f32(vel).reshape(-1, 3).copy()
    bpos = _f32(bpos).reshape(-1, 3)
    bparams = _f32(bparams).reshape(-1, 3)
    inv_mass = _f32(inv_mass)
    params = _f32(params).reshape(-1, 4)
    n = len(pos)
    dens = np.empty(n, np.float32)
    pres = np.empty(n, np.float32)
    forces = np.empty((n, 3), np.float32)
    rc = lib().orc_sph_step(_p(pos), _p(vel), _p(inv_mass), _p(params), C.c_int(n), _p(bpos),
                            _p(bparams), C.c_int(len(bpos)), C.c_float(grid), C.c_float(rate),
                            C.c_float(dt), _p(dens), _p(pres), _p(forces), C.c_int(nthreads))
    assert rc == 0
    return vel, dens, pres, forces


def stable_sort(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    perm = np.empty(len(keys), np.uint32)
    skeys = np.empty(len(keys), np.uint32)
    assert lib().orc_stable_sort(_p(keys), C.c_int(len(keys)), _p(perm), _p(skeys)) == 0
    return perm, skeys


def cell_table(sorted_keys, ncells):
    sorted_keys = np.ascontiguousarray(sorted_keys, dtype=np.uint32)
    table = np.empty((ncells, 2), np.int32)
    lib().orc_cell_table(_p(sorted_keys), C.c_int(len(sorted_keys)), _p(table), C.c_int(ncells))
    return table


def dense_keys(cells, origin, dims):
    cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
    origin = np.ascontiguousarray(origin, dtype=np.int32)
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    keys = np.empty(len(cells), np.uint32)
    lib().orc_dense_keys(_p(cells), C.c_int(len(cells)), _p(origin), _p(dims), _p(keys))
    return keys


def count_near_hash_collisions(lo, hi):
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    hi = np.ascontiguousarray(hi, dtype=np.int32)
    return int(lib().orc_count_near_hash_collisions(_p(lo), _p(hi)))


def integrate_particles(pos, vel, inv_mass, params, boxes, dt, particle_restitution=0.5):
    """Row I, particles.  boxes: (K, 25) float32 (SphBox layout).  Returns pos, vel, impulses."""
    pos = _f32(pos).reshape(-1, 3).copy()
    vel = _f32(vel).reshape(-1, 3).copy()
    inv_mass = _f32(inv_mass)
    params = _f32(params).reshape(-1, 4)
    boxes = _f32(boxes).reshape(-1, BOX_FLOATS)
    imp = np.zeros((max(len(boxes), 1), 6), np.float32)
    lib().orc_integrate_particles(_p(pos), _p(vel), _p(inv_mass), _p(params), C.c_int(len(pos)),
                                  _p(boxes), C.c_int(len(boxes)), C.c_float(dt),
                                  C.c_float(particle_restitution), _p(imp))
    return pos, vel, imp[:len(boxes)]


def integrate_bodies(boxes, impulses, dt, gravity=(0.0, -9.81, 0.0), bounds=None, contacts=False):
    boxes = _f32(boxes).reshape(-1, BOX_FLOATS).copy()
    imp = _f32(impulses).reshape(-1, 6)
    g = _f32(gravity)
    if bounds is None:
        bmin = bmax = np.zeros(3, np.float32)
        bounded = 0
    else:
        bmin, bmax = _f32(bounds[0]), _f32(bounds[1])
        bounded = 1
    lib().orc_integrate_bodies2(_p(boxes), C.c_int(len(boxes)), _p(imp), C.c_float(dt), _p(g),
                                C.c_int(bounded), _p(bmin), _p(bmax), C.c_int(1 if contacts else 0))
    return boxes


def sample_density(pos, inv_mass, params, origin, spacing, dims, rate=4.0):
    pos = _f32(pos).reshape(-1, 3)
    inv_mass = _f32(inv_mass)
    params = _f32(params).reshape(-1, 4)
    o = _f32(origin)
    d = np.ascontiguousarray(dims, np.int32)
    out = np.empty((int(d[2]), int(d[1]), int(d[0])), np.float32)
    lib().orc_sample_density(_p(pos), _p(inv_mass), _p(params), C.c_int(len(pos)), C.c_float(rate),
                             _p(o), C.c_float(spacing), _p(d), _p(out))
    return out


def max_threads():
    return int(lib().orc_max_threads())
