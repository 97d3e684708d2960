"""Synthetic scene generator: the reference's lattice seeding and the benchmark tanks.

Follows FluidInitializeSystem.CreateParticles (reference Assets/Scripts/FluidInitializeSystem.cs
:63-92): an authoring volume's local AABB is filled with a regular lattice at spacing 2*radius in
world units, x fastest, then y, then z; every lattice point is mapped through the volume's
local-to-world matrix.  Physical parameters are the prefab values (Assets/Prefabs/Particle.prefab
:244,292-295 and BoundaryParticle.prefab:97-99).  All arithmetic is float32, as in the reference.
"""
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32

# prefab values -------------------------------------------------------------------------------
RADIUS = F32(0.25)
MASS = F32(125.0)
REST_DENSITY = F32(1000.0)
VISCOSITY = F32(256.0)
GAS_CONSTANT = F32(512.0)
KERNEL_RADIUS_RATE = F32(4.0)  # FluidSPHUtils.cs:128
DT = F32(1.0) / F32(60.0)      # ConvertedSubScene.unity:279 (FPS 60)
BOX_FLOATS = 25                # SphBox, include/sphfluid.h


@dataclass
class Scene:
    pos: np.ndarray        # (N,3) f32  Translation.Value
    vel: np.ndarray        # (N,3) f32  PhysicsVelocity.Linear
    inv_mass: np.ndarray   # (N,)  f32  PhysicsMass.InverseMass
    params: np.ndarray     # (N,4) f32  FluidParticleComponent
    bpos: np.ndarray       # (B,3) f32
    bparams: np.ndarray    # (B,3) f32  BoundaryParticleComponent (radius, mass, restDensity)
    grid_size: float
    boxes: np.ndarray = field(default_factory=lambda: np.zeros((0, BOX_FLOATS), F32))
    tank_min: np.ndarray = None
    tank_max: np.ndarray = None
    name: str = ""

    @property
    def n(self):
        return len(self.pos)

    @property
    def b(self):
        return len(self.bpos)


def quat_matrix(q):
    x, y, z, w = [F32(v) for v in q]
    one, two = F32(1), F32(2)
    return np.array([
        [one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
        [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
        [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)]], dtype=F32)


def create_particles(position, rotation, scale, radius, bounds_min=(-0.5, -0.5, -0.5),
                     bounds_size=(1.0, 1.0, 1.0)):
    """CreateParticles (FluidInitializeSystem.cs:63-92) for a TRS local-to-world transform."""
    scale = np.asarray(scale, F32)
    rot = quat_matrix(rotation)
    l2w = rot * scale[None, :]            # columns = scaled axes
    radius = F32(radius)
    # radius / |l2w * axis|  (:65-67)
    rad = np.array([radius / F32(np.sqrt(np.sum(l2w[:, a].astype(F32) ** 2, dtype=F32)))
                    for a in range(3)], F32)
    size = np.asarray(bounds_size, F32)
    bmin = np.asarray(bounds_min, F32)
    num = [int((size[a] + rad[a]) / (F32(2) * rad[a])) for a in range(3)]   # (:69-71)
    idx = [np.arange(num[a], dtype=F32) for a in range(3)]
    # local lattice coordinate  x*2*xRadius + bounds.Min.x + xRadius  (:84-86)
    loc = [idx[a] * F32(2) * rad[a] + bmin[a] + rad[a] for a in range(3)]
    lz, ly, lx = np.meshgrid(loc[2], loc[1], loc[0], indexing="ij")   # x fastest (:73-77)
    local = np.stack([lx.ravel(), ly.ravel(), lz.ravel()], axis=1).astype(F32)
    pos = np.asarray(position, F32)
    world = np.empty_like(local)
    for r in range(3):  # MultiplyPoint3x4
        world[:, r] = (l2w[r, 0] * local[:, 0] + l2w[r, 1] * local[:, 1]) + \
                      l2w[r, 2] * local[:, 2] + pos[r]
    return world, tuple(num)


def _fluid_arrays(n, rest_density=REST_DENSITY):
    params = np.empty((n, 4), F32)
    params[:, 0] = RADIUS
    params[:, 1] = rest_density
    params[:, 2] = VISCOSITY
    params[:, 3] = GAS_CONSTANT
    inv_mass = np.full(n, F32(1.0) / MASS, F32)
    return params, inv_mass


def _boundary_params(b):
    bp = np.empty((b, 3), F32)
    bp[:, 0] = RADIUS
    bp[:, 1] = MASS
    bp[:, 2] = REST_DENSITY
    return bp


def static_box(center, half_extents, rotation=(0, 0, 0, 1), restitution=0.0, friction=0.0):
    b = np.zeros(BOX_FLOATS, F32)
    b[0:3] = center
    b[3:7] = rotation
    b[7:10] = half_extents
    b[20] = restitution
    b[21] = friction
    return b


def dynamic_box(center, half_extents, mass, rotation=(0, 0, 0, 1), restitution=0.0,
                friction=0.5, lin_damping=0.01, ang_damping=0.05, gravity_factor=1.0):
    b = static_box(center, half_extents, rotation, restitution, friction)
    hx, hy, hz = [float(v) for v in half_extents]
    b[16] = 1.0 / mass
    ixx = mass / 3.0 * (hy * hy + hz * hz)
    iyy = mass / 3.0 * (hx * hx + hz * hz)
    izz = mass / 3.0 * (hx * hx + hy * hy)
    b[17:20] = (1.0 / ixx, 1.0 / iyy, 1.0 / izz)
    b[22] = lin_damping
    b[23] = ang_damping
    b[24] = gravity_factor
    return b


def demo_scene(with_buoyancy_object=True):
    """C1: the shipped scene (ConvertedSubScene.unity): 4800 fluid + 4000 boundary particles.

    Water volume :666-667; walls :139-171 (ceiling), :292-320, :353-377, :410-434, :467-495,
    :528-556 (floor) over the Wall prefab's scale (10,10,0.5) (Wall.prefab:33);
    BuoyancyObject :704-712,849-860,943-944.
    """
    s = 0.7071068
    water, num = create_particles((-6, 5, 0), (0, 0, 0, 1), (6, 10, 10), RADIUS)
    walls = [
        ((0, 10, 0), (s, -0.0, -0.0, -s), (20, 10, 0.5)),   # ceiling
        ((0, 5, 5), (0, 0, 0, 1), (20, 10, 0.5)),
        ((10, 5, 0), (0, s, 0, s), (10, 10, 0.5)),
        ((-10, 5, 0), (0, s, -0.0, -s), (10, 10, 0.5)),
        ((0, 5, -5), (0, 1, 0, 0), (20, 10, 0.5)),
        ((0, 0, 0), (s, 0, 0, s), (20, 10, 0.5)),            # floor
    ]
    bparts, boxes = [], []
    for p, q, sc in walls:
        w, _ = create_particles(p, q, sc, RADIUS)
        bparts.append(w)
        boxes.append(static_box(p, [0.5 * v for v in sc], q))
    bpos = np.concatenate(bparts, axis=0)
    if with_buoyancy_object:
        boxes.append(dynamic_box((0, 6.86, 0), (1.0, 2.0, 1.0), 8377.0))
    params, inv_mass = _fluid_arrays(len(water))
    return Scene(pos=water, vel=np.zeros_like(water), inv_mass=inv_mass, params=params,
                 bpos=bpos, bparams=_boundary_params(len(bpos)),
                 grid_size=float(RADIUS * KERNEL_RADIUS_RATE), boxes=np.stack(boxes).astype(F32),
                 tank_min=np.array([-10, 0, -5], F32), tank_max=np.array([10, 10, 5], F32),
                 name="C1-demo")


def splitmix64_uniform(count, seed):
    """count uniform floats in [0,1) from splitmix64 (counter form: state_i = seed + (i+1)*G)."""
    g = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.arange(1, count + 1, dtype=np.uint64) * g)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(F32)


def jitter(pos, seed, amplitude=0.02):
    u = splitmix64_uniform(pos.size, seed).reshape(pos.shape)
    return (pos + (u * F32(2) - F32(1)) * F32(amplitude)).astype(F32)


def _plane(nu, nv, fixed_axis, fixed_value, spacing):
    u = (np.arange(nu, dtype=F32) + F32(0.5)) * spacing
    v = (np.arange(nv, dtype=F32) + F32(0.5)) * spacing
    vv, uu = np.meshgrid(v, u, indexing="ij")
    pts = np.empty((nu * nv, 3), F32)
    axes = [a for a in range(3) if a != fixed_axis]
    pts[:, axes[0]] = uu.ravel()
    pts[:, axes[1]] = vv.ravel()
    pts[:, fixed_axis] = fixed_value
    return pts


def dam_break(nx, ny, nz, jitter_seed=None, origin=(0.0, 0.0, 0.0), name=""):
    """SURVEY 8d tank: fluid block nx*ny*nz at spacing 0.5 filling the left third of a tank of
    interior (3nx, ny, nz)*0.5, one boundary-particle layer on each of the six faces, static wall
    boxes (0.5 thick) outside the faces."""
    sp = F32(2) * RADIUS
    ax = [(np.arange(k, dtype=F32) + F32(0.5)) * sp for k in (nx, ny, nz)]
    zz, yy, xx = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    pos = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], axis=1).astype(F32)
    tx, ty, tz = 3 * nx, ny, nz
    L = np.array([tx, ty, tz], F32) * sp
    faces = [
        _plane(ty, tz, 0, F32(0), sp), _plane(ty, tz, 0, L[0], sp),
        _plane(tx, tz, 1, F32(0), sp), _plane(tx, tz, 1, L[1], sp),
        _plane(tx, ty, 2, F32(0), sp), _plane(tx, ty, 2, L[2], sp),
    ]
    bpos = np.concatenate(faces, axis=0)
    org = np.asarray(origin, F32)
    pos += org
    bpos += org
    if jitter_seed is not None:
        pos = jitter(pos, jitter_seed)
    half = L * F32(0.5)
    t = F32(0.25)  # half thickness of a wall box
    boxes = []
    for a in range(3):
        for side in (0, 1):
            c = half.copy() + org
            c[a] = org[a] + (L[a] + t if side else -t)
            he = half.copy() + F32(2) * t
            he[a] = t
            boxes.append(static_box(c, he))
    params, inv_mass = _fluid_arrays(len(pos))
    return Scene(pos=pos, vel=np.zeros_like(pos), inv_mass=inv_mass, params=params, bpos=bpos,
                 bparams=_boundary_params(len(bpos)),
                 grid_size=float(RADIUS * KERNEL_RADIUS_RATE), boxes=np.stack(boxes).astype(F32),
                 tank_min=org.copy(), tank_max=org + L, name=name)


def splitmix64_at(index, seed):
    """splitmix64_uniform's value number `index` (0-based), for arbitrary index arrays."""
    g = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + ((np.asarray(index, np.uint64) + np.uint64(1)) * g)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(F32)


def dam_break_slab(nx, ny, nz, col0, col1, jitter_seed=None, margin_cols=8, amplitude=0.02,
                   name=""):
    """The columns col0 <= x-index < col1 of a dam break that runs along z: the fluid block
    nx*ny*nz (spacing 0.5) fills the first third of a tank of interior (nx, ny, 3nz)*0.5.  Used
    for the x-slab decomposition: every x column carries the same flow, so the slabs stay
    balanced.  Returns (Scene, global ids); ids and jitter depend only on the global lattice
    index, so any partition of the columns gives the same particles.  The boundary layers are
    cut to the columns within margin_cols of the slab; the wall boxes are the whole tank's."""
    sp = F32(2) * RADIUS
    xs = np.arange(col0, col1, dtype=np.int64)
    zz, yy, xx = np.meshgrid(np.arange(nz, dtype=np.int64), np.arange(ny, dtype=np.int64), xs,
                             indexing="ij")
    gid = ((zz * ny + yy) * nx + xx).ravel()
    pos = np.stack([(xx.ravel().astype(F32) + F32(0.5)) * sp,
                    (yy.ravel().astype(F32) + F32(0.5)) * sp,
                    (zz.ravel().astype(F32) + F32(0.5)) * sp], axis=1).astype(F32)
    if jitter_seed is not None:
        idx3 = (gid[:, None] * 3 + np.arange(3, dtype=np.int64)[None, :])
        u = splitmix64_at(idx3, jitter_seed)
        pos = (pos + (u * F32(2) - F32(1)) * F32(amplitude)).astype(F32)
    tx, ty, tz = nx, ny, 3 * nz
    L = np.array([tx, ty, tz], F32) * sp
    b0, b1 = max(0, col0 - margin_cols), min(tx, col1 + margin_cols)
    bx = (np.arange(b0, b1, dtype=F32) + F32(0.5)) * sp

    def plane_xu(nu, fixed_axis, fixed_value):   # spans x (cut) and one other axis
        other = 2 if fixed_axis == 1 else 1
        u = (np.arange(nu, dtype=F32) + F32(0.5)) * sp
        uu, xg = np.meshgrid(u, bx, indexing="ij")
        pts = np.empty((uu.size, 3), F32)
        pts[:, 0] = xg.ravel()
        pts[:, other] = uu.ravel()
        pts[:, fixed_axis] = fixed_value
        return pts

    faces = [plane_xu(tz, 1, F32(0)), plane_xu(tz, 1, L[1]),
             plane_xu(ty, 2, F32(0)), plane_xu(ty, 2, L[2])]
    if b0 == 0:
        faces.append(_plane(ty, tz, 0, F32(0), sp))
    if b1 == tx:
        faces.append(_plane(ty, tz, 0, L[0], sp))
    bpos = np.concatenate(faces, axis=0)
    half = L * F32(0.5)
    t = F32(0.25)
    boxes = []
    for a in range(3):
        for side in (0, 1):
            c = half.copy()
            c[a] = L[a] + t if side else -t
            he = half.copy() + F32(2) * t
            he[a] = t
            boxes.append(static_box(c, he))
    params, inv_mass = _fluid_arrays(len(pos))
    sc = Scene(pos=pos, vel=np.zeros_like(pos), inv_mass=inv_mass, params=params, bpos=bpos,
               bparams=_boundary_params(len(bpos)), grid_size=float(RADIUS * KERNEL_RADIUS_RATE),
               boxes=np.stack(boxes).astype(F32), tank_min=np.zeros(3, F32), tank_max=L,
               name=name)
    return sc, gid.astype(np.uint32)


def dam_break_columns(nx, ny, nz, col0, col1, jitter_seed=None, amplitude=0.02, name=""):
    """The fluid columns col0 <= x-index < col1 of dam_break(nx, ny, nz) -- the SURVEY 8d tank,
    water in the left third, flow along x, ACROSS the cuts of an x-slab decomposition -- with the
    whole tank's boundary layers and wall boxes (the cuts move with the flow, so every rank keeps
    them all).  Returns (Scene, global ids); ids and jitter depend only on the global lattice
    index: the union over any partition of the columns is dam_break(nx, ny, nz, jitter_seed)."""
    sp = F32(2) * RADIUS
    xs = np.arange(col0, col1, dtype=np.int64)
    zz, yy, xx = np.meshgrid(np.arange(nz, dtype=np.int64), np.arange(ny, dtype=np.int64), xs,
                             indexing="ij")
    gid = ((zz * ny + yy) * nx + xx).ravel()
    pos = np.stack([(xx.ravel().astype(F32) + F32(0.5)) * sp,
                    (yy.ravel().astype(F32) + F32(0.5)) * sp,
                    (zz.ravel().astype(F32) + F32(0.5)) * sp], axis=1).astype(F32)
    if jitter_seed is not None:
        idx3 = (gid[:, None] * 3 + np.arange(3, dtype=np.int64)[None, :])
        u = splitmix64_at(idx3, jitter_seed)
        pos = (pos + (u * F32(2) - F32(1)) * F32(amplitude)).astype(F32)
    tx, ty, tz = 3 * nx, ny, nz
    L = np.array([tx, ty, tz], F32) * sp
    faces = [
        _plane(ty, tz, 0, F32(0), sp), _plane(ty, tz, 0, L[0], sp),
        _plane(tx, tz, 1, F32(0), sp), _plane(tx, tz, 1, L[1], sp),
        _plane(tx, ty, 2, F32(0), sp), _plane(tx, ty, 2, L[2], sp),
    ]
    bpos = np.concatenate(faces, axis=0)
    half = L * F32(0.5)
    t = F32(0.25)
    boxes = []
    for a in range(3):
        for side in (0, 1):
            c = half.copy()
            c[a] = L[a] + t if side else -t
            he = half.copy() + F32(2) * t
            he[a] = t
            boxes.append(static_box(c, he))
    params, inv_mass = _fluid_arrays(len(pos))
    sc = Scene(pos=pos, vel=np.zeros_like(pos), inv_mass=inv_mass, params=params, bpos=bpos,
               bparams=_boundary_params(len(bpos)), grid_size=float(RADIUS * KERNEL_RADIUS_RATE),
               boxes=np.stack(boxes).astype(F32), tank_min=np.zeros(3, F32), tank_max=L, name=name)
    return sc, gid.astype(np.uint32)


CONFIGS = {
    # name: (nx, ny, nz, config index for the jitter seed)
    "C1": (12, 20, 20, 0),
    "C2": (64, 128, 128, 1),
    "C3": (256, 256, 256, 2),
    "C4": (1024, 512, 256, 3),
    "C5": (128, 128, 256, 4),
}
JITTER_SEED_BASE = 0x5EED0000


def make_config(name, jittered=True):
    """The BASELINE.json configurations.  C1 is the shipped demo scene."""
    if name == "C1":
        sc = demo_scene()
        if jittered:
            sc.pos = jitter(sc.pos, JITTER_SEED_BASE + 0)
        return sc
    nx, ny, nz, ci = CONFIGS[name]
    sc = dam_break(nx, ny, nz, jitter_seed=(JITTER_SEED_BASE + ci) if jittered else None,
                   name=name)
    if name == "C5":
        # non-uniform rest density: the x-upper half of the block is a lighter fluid
        mid = np.median(sc.pos[:, 0])
        sc.params[sc.pos[:, 0] > mid, 1] = F32(800.0)
        # 64 dynamic boxes (2 x 4 x 2, mass 8377 as the BuoyancyObject) on a seeded grid in the
        # dry two thirds of the tank, above where the free surface will settle
        u = splitmix64_uniform(64 * 2, JITTER_SEED_BASE + 0x100 + ci).reshape(64, 2)
        L = sc.tank_max - sc.tank_min
        dyn = []
        for k in range(64):
            gx, gz = k % 8, k // 8
            cx = sc.tank_min[0] + L[0] / 3.0 + (gx + 0.5) / 8.0 * (2.0 * L[0] / 3.0) \
                + (u[k, 0] - 0.5) * 0.5
            cz = sc.tank_min[2] + (gz + 0.5) / 8.0 * L[2] + (u[k, 1] - 0.5) * 0.5
            cy = sc.tank_min[1] + 0.6 * L[1]
            dyn.append(dynamic_box((cx, cy, cz), (1.0, 2.0, 1.0), 8377.0))
        sc.boxes = np.concatenate([sc.boxes, np.stack(dyn).astype(F32)], axis=0)
    return sc


def chunk_first_radii(params, chunk_capacity=128):
    """Radius of the first particle of every ECS chunk (what FindGridSizeJob reads,
    FluidSPHUtils.cs:35-43).  Chunk capacity is an ECS detail; any value gives the same grid size
    when radii are uniform."""
    return np.ascontiguousarray(params[::chunk_capacity, 0])
