"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/sphfluid.h
declares, struct layouts match the reference's components, and host-side helpers work.  No
compute call is made (there is no GPU here and the library has no CPU path)."""
import ctypes as C
import os
import re

import numpy as np

from fluidsimulation_amd import native, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sphfluid.h")).read()
    return sorted(set(re.findall(r"SPH_API\s+[\w\s\*]+?\b(sph_\w+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(native.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = native.load()
    for name in _declared():
        assert getattr(lib, name) is not None, name


def test_struct_layouts_match_the_reference_components():
    # FluidParticleComponent 4 x f32, BoundaryParticleComponent 3 x f32 (checked by
    # static_asserts in the library too); SphParams / SphGrid / SphTimings mirror the header
    assert C.sizeof(native.SphParams) == 4 * 15
    assert C.sizeof(native.SphGrid) == 4 * 8
    assert C.sizeof(native.SphConfig) == 16
    assert C.sizeof(native.SphTimings) == 9 * 8 + 9 * 8 + 8
    assert scene.BOX_FLOATS * 4 == 100


def test_find_grid_size_matches_oracle(oracle):
    rng = np.random.default_rng(1)
    params = rng.uniform(0.1, 1.0, (1000, 4)).astype(np.float32)
    first = np.arange(0, 1000, 128, dtype=np.int32)
    got = native.find_grid_size(params, first, 4.0)
    assert got == oracle.find_grid_size(params[first, 0], 4.0)
    assert native.find_grid_size(params[:0], first[:0], 4.0) == np.float32(3.4028235e38)


def test_create_fails_loudly_without_a_gpu():
    """no CPU fallback: on a machine without a HIP device sph_create returns SPH_E_NODEVICE"""
    lib = native.load()
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    cfg = native.SphConfig(0, 0, 0, 0)
    rc = lib.sph_create(C.byref(cfg), C.byref(h))
    assert rc == native.SPH_E_NODEVICE and not h
    assert b"no HIP device" in lib.sph_last_error(None)
