"""GPU parity tests, stage by stage, through the C ABI (libsphfluid.so) against the CPU oracle.

Bars (BASELINE.json north_star): hashes, sorted permutation and neighbour counts bit-exact;
densities, forces, velocities, positions within 1e-5 relative in fp32.
"""
import numpy as np
import pytest

from fluidsimulation_amd import native, scene
from tests.util import assert_close, jittered_block, load_scene, run_to_force

pytestmark = pytest.mark.gpu


def _scenes():
    yield scene.demo_scene()
    yield jittered_block(16, 16, 16, seed=0x5EED0003)


@pytest.mark.parametrize("which", [0, 1])
def test_hash_position(ctx, oracle, which):
    sc = list(_scenes())[which]
    p = load_scene(ctx, sc)
    ctx.hash_position(p)
    cells = oracle.quantize(sc.pos, sc.grid_size)
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_CELLS), cells)
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_HASHES), oracle.hash_cells(cells))
    origin, dims, bits, ncells = ctx.get_grid()
    assert ncells == dims[0] * dims[1] * dims[2] and (1 << bits) >= ncells
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_KEYS),
                                  oracle.dense_keys(cells, origin, dims))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 100_003, 1 << 20])
def test_radix_sort_random_keys(ctx, oracle, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    perm, skeys = ctx.sort_pairs(keys, 32)
    operm, okeys = oracle.stable_sort(keys)
    np.testing.assert_array_equal(skeys, okeys)
    np.testing.assert_array_equal(perm, operm)


@pytest.mark.parametrize("bits,hi", [(8, 7), (11, 1500), (24, 1 << 24), (32, 3)])
def test_radix_sort_many_ties(ctx, oracle, bits, hi):
    """few distinct keys -> long runs of ties: stability (ties by id) is what is being checked"""
    rng = np.random.default_rng(bits)
    keys = rng.integers(0, hi, size=300_000, dtype=np.uint64).astype(np.uint32)
    perm, skeys = ctx.sort_pairs(keys, bits)
    operm, okeys = oracle.stable_sort(keys)
    np.testing.assert_array_equal(skeys, okeys)
    np.testing.assert_array_equal(perm, operm)


def test_radix_sort_fnv_hash_keys(ctx, oracle):
    """the reference's own key (FNV-1 of the cell) sorts to the stable (hash, id) order"""
    sc = scene.demo_scene()
    h = oracle.hash_cells(oracle.quantize(sc.pos, sc.grid_size))
    perm, skeys = ctx.sort_pairs(h, 32)
    operm, okeys = oracle.stable_sort(h)
    np.testing.assert_array_equal(perm, operm)
    np.testing.assert_array_equal(skeys, okeys)


@pytest.mark.parametrize("which", [0, 1])
def test_sort_table_reorder(ctx, oracle, which):
    sc = list(_scenes())[which]
    p = load_scene(ctx, sc)
    ctx.hash_position(p)
    keys = ctx.read(native.SPH_BUF_KEYS)
    ctx.radix_sort()
    ctx.map_table()
    ctx.reorder()
    operm, okeys = oracle.stable_sort(keys)
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_SORTED_KEYS), okeys)
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_PERM), operm)
    ncells = ctx.get_grid()[3]
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_TABLE), oracle.cell_table(okeys, ncells))
    # boundary: sorted once with the same key
    origin, dims, _, _ = ctx.get_grid()
    bkeys = oracle.dense_keys(oracle.quantize(sc.bpos, sc.grid_size), origin, dims)
    bperm, bsorted = oracle.stable_sort(bkeys)
    np.testing.assert_array_equal(ctx.read_boundary_perm(sc.b), bperm)
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_BOUNDARY_TABLE),
                                  oracle.cell_table(bsorted, ncells))


# debug flag sets: parity outputs on (exact-predicate kernels), production kernels (branch-free),
# and the global-memory path the kernels take when a tile's halo overflows the LDS tile
_ALL = native.SPH_DEBUG_HASHES | native.SPH_DEBUG_COUNTS
_FLAGS = [_ALL, 0, _ALL | native.SPH_DEBUG_NO_LDS_TILE, native.SPH_DEBUG_NO_LDS_TILE]


@pytest.mark.parametrize("flags", _FLAGS)
@pytest.mark.parametrize("which", [0, 1])
def test_density_pressure_force(ctx, oracle, which, flags):
    sc = list(_scenes())[which]
    if which == 1:
        rng = np.random.default_rng(7)
        sc.vel = rng.normal(0, 1.0, sc.pos.shape).astype(np.float32)
    p = load_scene(ctx, sc, debug=flags)
    run_to_force(ctx, p)
    dens, pres, cnt = oracle.density_pressure(sc.pos, sc.inv_mass, sc.params, sc.bpos,
                                              sc.grid_size)
    if flags & native.SPH_DEBUG_COUNTS:
        np.testing.assert_array_equal(ctx.read(native.SPH_BUF_DENSITY_COUNTS), cnt)
    gd = ctx.read(native.SPH_BUF_DENSITIES)
    gp = ctx.read(native.SPH_BUF_PRESSURES)
    assert_close(gd, dens, what="density")
    assert_close(gp, pres, what="pressure")
    # force parity is checked on the oracle's own densities AND end to end
    forces, fcnt = oracle.force(sc.pos, sc.inv_mass, sc.vel, sc.params, sc.bpos, sc.bparams,
                                dens, pres, sc.grid_size)
    if flags & native.SPH_DEBUG_COUNTS:
        np.testing.assert_array_equal(ctx.read(native.SPH_BUF_FORCE_COUNTS), fcnt)
    assert_close(ctx.read(native.SPH_BUF_FORCES), forces, rtol=2e-5, what="force")


def test_apply_force_and_integrate(ctx, oracle):
    sc = scene.demo_scene()
    rng = np.random.default_rng(11)
    sc.vel = rng.normal(0, 2.0, sc.pos.shape).astype(np.float32)
    p = load_scene(ctx, sc, colliders=True)
    run_to_force(ctx, p)
    dens = ctx.read(native.SPH_BUF_DENSITIES)
    forces = ctx.read(native.SPH_BUF_FORCES)
    ctx.apply_force(p)
    # ApplyForceJob is bit-exact given the same forces and densities
    vel = oracle.apply_force(sc.vel, forces, dens, sc.inv_mass, float(p.dt))
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_VELOCITIES), vel)
    ctx.integrate(p)
    opos, ovel, oimp = oracle.integrate_particles(sc.pos, vel, sc.inv_mass, sc.params, sc.boxes,
                                                  float(p.dt))
    gpos, gvel = ctx.download()
    assert_close(gpos, opos, what="position")
    assert_close(gvel, ovel, what="velocity")
    oboxes = oracle.integrate_bodies(sc.boxes, oimp, float(p.dt))
    assert_close(ctx.get_colliders(), oboxes, what="bodies")


def test_fast_particles_do_not_tunnel(ctx, oracle):
    """row I, swept: speeds of 1 to 3 wall thicknesses per step (|v| dt up to ~2.5; the discrete
    test of round 2 let such particles through the 0.5-thick walls) against the oracle, and every
    particle still inside the tank"""
    sc = scene.demo_scene()
    rng = np.random.default_rng(5)
    sc.vel = rng.normal(0, 40.0, sc.pos.shape).astype(np.float32)
    p = load_scene(ctx, sc, debug=0, colliders=True)
    ctx.integrate(p)
    opos, ovel, oimp = oracle.integrate_particles(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.boxes,
                                                  float(p.dt))
    gpos, gvel = ctx.download()
    speed = np.linalg.norm(sc.vel, axis=1) * float(p.dt)
    assert (speed > 0.5).mean() > 0.5 and speed.max() > 2.0
    assert_close(gpos, opos, what="position")
    assert_close(gvel, ovel, what="velocity")
    # the walls' mid-planes carry the boundary layer; inner faces 0.25 inside, centres a radius more
    lo = sc.bpos.min(0) + 0.5 - 1e-3
    hi = sc.bpos.max(0) - 0.5 + 1e-3
    assert ((gpos > lo) & (gpos < hi)).all()
    assert ((opos > lo) & (opos < hi)).all()
    oboxes = oracle.integrate_bodies(sc.boxes, oimp, float(p.dt))
    assert_close(ctx.get_colliders(), oboxes, rtol=1e-4, what="bodies")


def test_seed_volume_matches_create_particles(ctx):
    """sph_seed_volume = FluidInitializeSystem.CreateParticles (:63-92) on the device: positions
    bit-exact against the host generator for the demo's water volume and a rotated, scaled wall;
    a second volume appends with continuing ids"""
    s = 0.7071068
    vols = [((-6, 5, 0), (0, 0, 0, 1), (6, 10, 10)),           # water, ConvertedSubScene.unity:666-667
            ((0, 10, 0), (s, -0.0, -0.0, -s), (20, 10, 0.5))]  # ceiling wall, :139-171
    prm = np.array([0.25, 1000.0, 256.0, 512.0], np.float32)
    want = []
    for k, (pos, rot, scale) in enumerate(vols):
        w, num = scene.create_particles(pos, rot, scale, scene.RADIUS)
        want.append(w)
        l2w = np.zeros((3, 4), np.float32)
        l2w[:, :3] = scene.quat_matrix(rot) * np.asarray(scale, np.float32)[None, :]
        l2w[:, 3] = pos
        added = ctx.seed_volume(l2w, (-0.5, -0.5, -0.5), (1.0, 1.0, 1.0), 0.25, prm, 1.0 / 125.0,
                                1.0, append=k > 0)
        assert added == len(w) == num[0] * num[1] * num[2]
    want = np.concatenate(want)
    assert ctx.n == len(want) == 4800 + 800
    gpos, gvel = ctx.download()
    assert np.array_equal(gpos, want)
    assert not gvel.any()
    p = native.make_params(1.0)
    ctx.step_resident(p, 1)
    gpos, _ = ctx.download()
    assert np.isfinite(gpos).all()
    # seeded state == uploaded state: one step from each gives the same bits
    with native.Context(device=0) as other:
        params, inv_mass = scene._fluid_arrays(len(want))
        other.upload(want, np.zeros_like(want), inv_mass, params, 1.0)
        other.step_resident(p, 1)
        opos, _ = other.download()
    assert np.array_equal(gpos, opos)


def test_density_field_matches_oracle(ctx, oracle):
    """sph_sample_density (the reference's open roadmap item, readme_zh.md:303): the field on a
    25 x 21 x 23 grid through the shipped scene after three steps (the call sorts the moved state
    itself) against the oracle's brute-force sum, 1e-5 of the largest value; then with two
    materials"""
    sc = scene.demo_scene()
    p = load_scene(ctx, sc, debug=0, colliders=True)
    ctx.step_resident(p, 3)
    pos, _ = ctx.download()
    lo, hi = pos.min(axis=0) - 1.3, pos.max(axis=0) + 1.3
    dims = (25, 21, 23)
    spacing = float(((hi - lo) / (np.array(dims) - 1)).max())
    got = ctx.sample_density(p, lo, spacing, dims)
    want = oracle.sample_density(pos, sc.inv_mass, sc.params, lo, spacing, dims)
    assert want.max() > 900 and (want == 0).any()
    assert_close(got, want, what="density field")
    sc2 = jittered_block(8, 8, 8, seed=3)
    sc2.params[::2, 1] = np.float32(800.0)
    sc2.inv_mass[::2] = np.float32(1.0 / 100.0)
    p2 = load_scene(ctx, sc2, debug=0)
    lo2 = sc2.pos.min(axis=0) - 0.5
    got2 = ctx.sample_density(p2, lo2, 0.37, (16, 16, 16))
    want2 = oracle.sample_density(sc2.pos, sc2.inv_mass, sc2.params, lo2, 0.37, (16, 16, 16))
    assert_close(got2, want2, what="density field, two materials")
