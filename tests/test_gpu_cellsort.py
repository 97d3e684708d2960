"""The step entry points sort by counting per cell (sph_sort.h); the stage calls keep the stable
radix sort the oracle is compared with.  The order inside a cell that the counting sort leaves is
arbitrary (arrival order of atomics) and sph_reorder ranks every cell by particle id, so the two
sorts must give the SAME resident state, bit for bit, step after step: SPHFLUID_STEP_SORT=radix
selects the radix passes inside the step entry points for exactly this comparison."""
import os

import numpy as np
import pytest

from fluidsimulation_amd import native, scene
from tests.util import jittered_block, load_scene

pytestmark = pytest.mark.gpu


def run(sc, steps, radix, frames=()):
    if radix:
        os.environ["SPHFLUID_STEP_SORT"] = "radix"
    try:
        with native.Context(device=0) as c:
            p = load_scene(c, sc, debug=0, colliders=True)
            out = []
            done = 0
            for f in list(frames) + [steps]:
                c.step_resident(p, f - done)
                done = f
                out.append(c.download())
            return out
    finally:
        os.environ.pop("SPHFLUID_STEP_SORT", None)


@pytest.mark.parametrize("name", ["jitter", "demo", "dense"])
def test_counting_sort_and_radix_sort_give_the_same_state(name):
    if name == "jitter":
        sc, steps = jittered_block(20, 18, 16, seed=11), 30
    elif name == "demo":
        sc, steps = scene.make_config("C1"), 40
        # (without the demo's dynamic box: the pushes of the particles on a body are summed with
        # float atomics, in whatever order the waves arrive, with either sort)
        sc.boxes = sc.boxes[:6]
    else:  # eight times the rest density, 64 particles per cell: the crowded-cell path and its masks
        sc, steps = jittered_block(16, 16, 16, seed=5), 6
        sc.pos = (sc.pos * np.float32(0.5) + np.float32(0.25)).astype(np.float32)
    a = run(sc, steps, radix=False, frames=(1, steps // 2))
    b = run(sc, steps, radix=True, frames=(1, steps // 2))
    for (pa, va), (pb, vb) in zip(a, b):
        assert np.isfinite(pa).all()
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(va, vb)


def test_step_and_compat_use_it_too(oracle):
    """sph_step_compat through the counting sort against the oracle's density and force"""
    sc = jittered_block(12, 12, 12, seed=4)
    with native.Context(device=0) as c:
        c.set_boundary(sc.bpos, sc.bparams)
        p = native.make_params(sc.grid_size)
        dens, forces = c.step_compat(p, sc.pos, sc.vel, sc.inv_mass, sc.params)
    d0, p0, _ = oracle.density_pressure(sc.pos, sc.inv_mass, sc.params, sc.bpos, 1.0)
    f0, _ = oracle.force(sc.pos, sc.inv_mass, sc.vel, sc.params, sc.bpos, sc.bparams, d0, p0, 1.0)
    np.testing.assert_allclose(dens, d0, rtol=2e-6)
    scale = np.abs(f0).max()
    np.testing.assert_allclose(forces, f0, rtol=2e-5, atol=2e-5 * scale)
