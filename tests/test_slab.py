"""Slab decomposition: world_size-2 (and 3) runs over gloo against the undecomposed
run.  The CPU test drives fluidsimulation_amd.slab.SlabSimulation with the numpy/oracle backend;
the GPU test drives it with libsphfluid's sph_slab_* entry points, both ranks on device 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(backend, world, nsteps, tmp_path, bodies=False, timeout=600):
    out = tmp_path / f"verdict_{backend}_{world}.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "slab_worker.py"), backend, str(nsteps),
           str(out), "1" if bodies else "0"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(out.read_text())


def check(v, rtol):
    assert v["ok"], v
    assert v["gathered"] == v["n"] == v["unique_ids"]          # every particle has one owner
    assert all(s > 0 for s in v["sent_records"])                # every rank sent ghosts
    assert v["moved_owner"] > 0                                  # ownership did change hands
    assert v["max_pos_err"] <= rtol and v["max_vel_err"] <= rtol and v["max_rho_err"] <= rtol, v


def check_bodies(v):
    assert v["body_moved"] > 1e-3          # the box was pushed
    assert v["max_body_err"] <= 1e-4, v    # and every rank's copy agrees with the one-domain run


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchange_on_cpu_ranks(world, tmp_path):
    """positions, velocities, densities equal the single-domain oracle run to 1e-5 after 4 steps
    with particles crossing the slab faces"""
    check(run_ranks("numpy", world, 4, tmp_path), 1e-5)


def test_slab_rigid_body_impulses_are_summed_over_cpu_ranks(tmp_path):
    """a dynamic box straddling the slab face is pushed by both ranks' particles"""
    v = run_ranks("numpy", 2, 4, tmp_path, bodies=True)
    check(v, 1e-5)
    check_bodies(v)


@pytest.mark.gpu
def test_slab_exchange_on_the_gpu(tmp_path):
    """the same through sph_slab_pack / sph_slab_import, two ranks sharing device 0"""
    check(run_ranks("hip", 2, 4, tmp_path), 1e-5)


@pytest.mark.gpu
def test_slab_rigid_body_on_the_gpu(tmp_path):
    v = run_ranks("hip", 2, 4, tmp_path, bodies=True)
    check(v, 1e-5)
    check_bodies(v)
