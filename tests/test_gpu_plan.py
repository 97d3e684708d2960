"""The neighbour plan's memory: a scene that needs the most plan per particle (one particle per
cell), and a pool that is too small, which must end as SPH_E_TOOLARGE and never as a GPU fault
(an entry the plan could not fill used to be left unwritten in the block list)."""
import os

import numpy as np
import pytest

from fluidsimulation_amd import native, scene
from tests.util import assert_close, assert_close_elementwise, jittered_block, load_scene, run_to_force

pytestmark = pytest.mark.gpu


def test_one_particle_per_cell(ctx, oracle):
    """lattice spacing = cell size: every cell holds one target, a tile's plan is at its largest
    per particle (a unit, a mask word of 64 lanes and nine windows each)"""
    sc = scene.dam_break(2, 2, 2)
    g = np.arange(20, dtype=np.float32) + np.float32(0.5)
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pos = scene.jitter(pos, 7, 0.3)
    n = len(pos)
    sc.pos, sc.vel = pos, np.zeros_like(pos)
    sc.params = np.tile(sc.params[:1], (n, 1))
    sc.inv_mass = np.tile(sc.inv_mass[:1], n)
    sc.bpos, sc.bparams = sc.bpos[:0], sc.bparams[:0]
    p = load_scene(ctx, sc, debug=0)
    run_to_force(ctx, p)
    ctx.synchronize()  # reports a plan overflow
    dens, pres, cnt = oracle.density_pressure(sc.pos, sc.inv_mass, sc.params, sc.bpos, 1.0)
    forces, fcnt = oracle.force(sc.pos, sc.inv_mass, sc.vel, sc.params, sc.bpos, sc.bparams, dens,
                                pres, 1.0)
    assert_close_elementwise(ctx.read(native.SPH_BUF_DENSITIES), dens, what="density")
    assert_close(ctx.read(native.SPH_BUF_FORCES), forces, rtol=2e-5, what="force")
    np.testing.assert_array_equal(ctx.read(native.SPH_BUF_MASK_COUNTS), cnt)


def test_plan_pool_overflow_is_reported_not_a_fault():
    sc = jittered_block(16, 16, 16, seed=3)
    os.environ["SPHFLUID_POOL_LIMIT"] = "12000"  # 16-byte units: room for a few blocks only
    try:
        with native.Context(device=0) as c:
            p = load_scene(c, sc, debug=0, colliders=True)
            c.step_resident(p, 2)   # runs on: blocks without a plan are skipped
            with pytest.raises(native.SphError) as e:
                c.download()
            assert e.value.code == native.SPH_E_TOOLARGE
    finally:
        del os.environ["SPHFLUID_POOL_LIMIT"]
    with native.Context(device=0) as c:  # and the same scene with the pool it asks for
        p = load_scene(c, sc, debug=0, colliders=True)
        c.step_resident(p, 2)
        pos, vel = c.download()
        assert np.isfinite(pos).all() and np.isfinite(vel).all()


def test_crowded_cells_without_room_for_their_masks_walk_every_candidate():
    """The hit masks of a crowded cell come out of the plan pool; when it is full the cell simply
    has none and its force pass walks every candidate (no error): same forces, and the audit
    buffer says -1 for exactly those particles."""
    sc = jittered_block(20, 20, 20, seed=9)
    sc.pos = (sc.pos * np.float32(0.5) + np.float32(0.25)).astype(np.float32)  # 64 per cell

    def forces(limit):
        if limit:
            os.environ["SPHFLUID_POOL_LIMIT"] = str(limit)
        try:
            with native.Context(device=0) as c:
                p = load_scene(c, sc, debug=0)
                run_to_force(c, p)
                c.synchronize()
                return c.read(native.SPH_BUF_FORCES), c.read(native.SPH_BUF_MASK_COUNTS)
        finally:
            os.environ.pop("SPHFLUID_POOL_LIMIT", None)

    f_all, cnt_all = forces(0)
    # (every plan workgroup takes 128 KB pieces of the pool first: room for those, little more)
    f_some, cnt_some = forces(140000)
    assert (cnt_all >= 0).all()
    nomask = cnt_some < 0
    assert 0 < nomask.sum() < len(sc.pos), f"{nomask.sum()} particles without masks"
    np.testing.assert_array_equal(cnt_some[~nomask], cnt_all[~nomask])
    assert_close(f_some, f_all, rtol=1e-6, what="force without masks")
