"""Slab decomposition: world_size-2 (and 3) runs over gloo against the undecomposed
run.  The CPU test drives fluidsimulation_amd.slab.SlabSimulation with the numpy/oracle backend;
the GPU test drives it with libsphfluid's sph_slab_* entry points, both ranks on device 0."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch  # noqa: F401  (before libsphfluid: see fluidsimulation_amd/native.py load())

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(backend, world, nsteps, tmp_path, bodies=False, rebalance=False, timeout=600):
    out = tmp_path / f"verdict_{backend}_{world}.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "slab_worker.py"), backend, str(nsteps),
           str(out), "1" if bodies else "0", "1" if rebalance else "0"]
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


def test_slab_cuts_move_towards_balance_on_cpu_ranks(tmp_path):
    """cuts that start with most of the fluid on rank 0 move, one halo-limited shift per step,
    towards equal counts; the run still equals the undecomposed one"""
    v = run_ranks("numpy", 3, 6, tmp_path, rebalance=True)
    check(v, 1e-5)
    owned = v["owned_per_rank"]
    first = [o[0] for o in owned]
    last = [o[-1] for o in owned]
    assert max(first) - min(first) > max(last) - min(last), (first, last)
    assert last[0] < first[0]


@pytest.mark.gpu
def test_slab_exchange_on_the_gpu(tmp_path):
    """the same through sph_slab_pack / sph_slab_import, two ranks sharing device 0"""
    check(run_ranks("hip", 2, 4, tmp_path), 1e-5)


@pytest.mark.gpu
def test_slab_middle_rank_on_the_gpu(tmp_path):
    """three ranks on device 0: the middle one packs, sends and imports on both faces"""
    check(run_ranks("hip", 3, 4, tmp_path), 1e-5)


@pytest.mark.gpu
def test_slab_rigid_body_on_the_gpu(tmp_path):
    v = run_ranks("hip", 2, 4, tmp_path, bodies=True)
    check(v, 1e-5)
    check_bodies(v)


def test_slab_scene_is_partition_independent():
    """dam_break_slab: ids and jittered positions depend only on the global lattice index, so any
    split of the x columns yields the same particles (what lets every rank build its own slab)"""
    import numpy as np
    from fluidsimulation_amd import scene
    whole, wid = scene.dam_break_slab(16, 6, 5, 0, 16, jitter_seed=5)
    by_id = dict(zip(wid.tolist(), map(tuple, whole.pos.tolist())))
    assert len(by_id) == whole.n == 16 * 6 * 5
    seen = 0
    for lo, hi in ((0, 5), (5, 12), (12, 16)):
        part, pid = scene.dam_break_slab(16, 6, 5, lo, hi, jitter_seed=5)
        assert all(by_id[k] == tuple(v) for k, v in zip(pid.tolist(), part.pos.tolist()))
        assert np.all((part.pos[:, 0] >= lo * 0.5 - 0.02) & (part.pos[:, 0] < hi * 0.5 + 0.02))
        seen += part.n
    assert seen == whole.n


def test_material_table_is_sorted_union():
    import numpy as np
    from fluidsimulation_amd import slab
    params = np.array([[0.25, 1000, 256, 512]] * 3 + [[0.25, 800, 256, 512]] * 2, np.float32)
    inv_mass = np.full(5, 1 / 125, np.float32)
    table, index = slab.material_table(inv_mass, params)
    assert table.shape == (2, 5) and index.tolist() in ([0, 0, 0, 1, 1], [1, 1, 1, 0, 0])
    assert np.array_equal(table[index][:, :4], params)


@pytest.mark.gpu
def test_slab_error_paths():
    import numpy as np
    from fluidsimulation_amd import native, scene
    sc = scene.dam_break(4, 4, 4)
    ctx = native.Context(device=0)
    with pytest.raises(native.SphError):           # pack before upload
        ctx.slab_pack(0.0, 1.0, 2.0, 0, 0, 0)
    mats = np.array([[0.25, 1000, 256, 512, 1 / 125]], np.float32)
    idx = np.zeros(sc.n, np.uint32)
    ids = np.arange(sc.n, dtype=np.uint32)
    with pytest.raises(native.SphError):           # capacity below N
        ctx.slab_upload(sc.pos, sc.vel, idx, ids, mats, sc.grid_size, capacity=sc.n - 1)
    with pytest.raises(native.SphError):           # material index out of range
        ctx.slab_upload(sc.pos, sc.vel, idx + 1, ids, mats, sc.grid_size, capacity=2 * sc.n)
    ctx.slab_upload(sc.pos, sc.vel, idx, ids, mats, sc.grid_size, capacity=sc.n + 4)
    assert ctx.slab_counts() == {"resident": sc.n, "owned": sc.n, "ghosts": 0, "dead": 0}
    with pytest.raises(native.SphError):           # ECS-ordered views are refused in slab mode
        ctx.download()
    import torch
    rec = torch.zeros((8, 8), dtype=torch.float32, device="cuda:0")
    with pytest.raises(native.SphError) as e:       # a send buffer that is too small: reported
        ctx.slab_pack(1.0, 1.5, 2.0, rec.data_ptr(), rec.data_ptr(), 4)   # with nothing changed
    assert e.value.code == native.SPH_E_TOOLARGE
    need = ctx.last_slab_pack_counts
    assert need[0] > 4 and need[1] > 4 and need[2] + need[3] > 0
    assert ctx.slab_counts() == {"resident": sc.n, "owned": sc.n, "ghosts": 0, "dead": 0}
    # more records than the capacity holds: the buffers grow and keep the resident records
    rec[:, 0:3] = torch.tensor([0.3, 0.3, 0.3])
    rec[:, 7] = (torch.arange(8, dtype=torch.int32) + ((0x3FFF0000 | (1 << 31)) - (1 << 32))).view(torch.float32)
    torch.cuda.synchronize()
    ctx.slab_import(rec.data_ptr(), 8, 0)
    assert ctx.slab_counts() == {"resident": sc.n + 8, "owned": sc.n, "ghosts": 8, "dead": 0}
    # everything within the halo of an open-ended slab: nothing to send, nothing moves
    rec2 = torch.zeros((8, 8), dtype=torch.float32, device="cuda:0")
    assert ctx.slab_pack(-np.inf, np.inf, 2.0, rec2.data_ptr(), rec2.data_ptr(), 8) == [0, 0, 0, 0]
    p = native.make_params(sc.grid_size)
    ctx.step_resident(p, 2)
    pos, vel, rho, pid = ctx.slab_download()
    assert len(pid) == sc.n and np.array_equal(np.sort(pid), ids) and np.isfinite(pos).all()
    ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)   # leaves slab mode
    ctx.step_resident(p, 1)
    ctx.download()
    ctx.close()


def test_plan_cuts_is_bounded_and_converges():
    """plan_cuts: moves towards equal counts, never by more than max_shift, never beyond one halo
    inside the old neighbouring slabs, interior slabs stay one halo wide"""
    import math
    from fluidsimulation_amd.slab import plan_cuts
    halo, dom = 2.0, (0.0, 40.0)
    edges = [-math.inf, 30.0, 34.0, math.inf]
    density = lambda lo, hi: 100.0 * (min(hi, 40.0) - max(lo, 0.0))   # uniform fluid on [0, 40]
    for _ in range(60):
        counts = [density(edges[r] if r else 0.0, edges[r + 1] if r < 2 else 40.0) for r in range(3)]
        new = plan_cuts(edges, counts, halo, 1.5, dom)
        for f in (1, 2):
            assert abs(new[f] - edges[f]) <= 1.5 + 1e-9
        assert new[1] <= edges[2] - halo + 1e-9 and new[2] >= edges[1] + halo - 1e-9
        assert new[2] - new[1] >= halo - 1e-9
        edges = new
    assert abs(edges[1] - 40.0 / 3) < 0.5 and abs(edges[2] - 80.0 / 3) < 0.5
    assert plan_cuts([-math.inf, 1.0, math.inf], [0, 0], halo, 1.0) == [-math.inf, 1.0, math.inf]


@pytest.mark.gpu
def test_slab_cuts_move_on_the_gpu(tmp_path):
    v = run_ranks("hip", 2, 5, tmp_path, rebalance=True)
    check(v, 1e-5)
    assert v["final_edges"][0] < 4.6     # the cut started at 60 % of the block and moved left


@pytest.mark.gpu
def test_native_rccl_exchange_world_of_one():
    """sph_slab_comm_init / sph_slab_exchange through the C ABI with a real RCCL communicator of
    one rank (the GPU box has one device; RCCL refuses two ranks on one GPU): the unique id, the
    communicator, the count / pack / import sequence on the library's stream, the bookkeeping, and
    that a run driven by sph_slab_exchange equals the run driven by sph_slab_pack bit for bit.  The
    two-neighbour message pattern itself is what the gloo tests above rehearse rank by rank."""
    import numpy as np
    import torch
    from fluidsimulation_amd import native, scene
    sc = scene.dam_break(6, 6, 6, jitter_seed=11)
    mats = np.array([[0.25, 1000, 256, 512, 1 / 125]], np.float32)
    idx = np.zeros(sc.n, np.uint32)
    ids = np.arange(sc.n, dtype=np.uint32)
    p = native.make_params(sc.grid_size)
    uid = native.Context.slab_comm_unique_id()
    assert len(uid) == 128
    out = []
    for use_native in (True, False):
        with native.Context(device=0) as ctx:
            ctx.set_boundary(sc.bpos, sc.bparams)
            ctx.slab_upload(sc.pos, sc.vel, idx, ids, mats, sc.grid_size, capacity=2 * sc.n)
            if use_native:
                with pytest.raises(native.SphError):   # exchange before the communicator exists
                    ctx.slab_exchange(-np.inf, np.inf, 2.0)
                ctx.slab_comm_init(0, 1, uid)
                with pytest.raises(native.SphError):   # a second communicator on one context
                    ctx.slab_comm_init(0, 1, uid)
            rec = torch.zeros((16, 8), dtype=torch.float32, device="cuda:0")
            for _ in range(3):
                if use_native:
                    assert ctx.slab_exchange(-np.inf, np.inf, 2.0) == [0, 0, 0, 0]
                else:
                    assert ctx.slab_pack(-np.inf, np.inf, 2.0, rec.data_ptr(), rec.data_ptr(), 8) == [0, 0, 0, 0]
                ctx.step_resident(p, 1)
            assert ctx.slab_counts() == {"resident": sc.n, "owned": sc.n, "ghosts": 0, "dead": 0}
            out.append(ctx.slab_download())
            if use_native:
                ctx.slab_comm_destroy()
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_counts_of_the_next_exchange_come_from_the_integrate_kernel():
    """With the cuts of the last exchange unchanged, the counters of the next one (records and
    transfers per side) are accumulated by the integrate kernel and the counting pass of
    k_slab_pack is skipped; SPHFLUID_SLAB_COUNT_PASS=1 forces that pass.  Cuts inside the fluid,
    nobody to send to: what the pack reports and what stays resident must be the same step by
    step, also when the cuts change in between (then the pass of its own runs)."""
    import os
    import numpy as np
    import torch
    from fluidsimulation_amd import native, scene
    sc = scene.dam_break(12, 10, 8, jitter_seed=3)
    mats = np.array([[0.25, 1000, 256, 512, 1 / 125]], np.float32)
    idx = np.zeros(sc.n, np.uint32)
    ids = np.arange(sc.n, dtype=np.uint32)
    p = native.make_params(sc.grid_size)
    cuts = [(2.0, 4.0)] * 6 + [(2.5, 4.5)] * 3 + [(2.0, 4.0)] * 6
    runs = []
    for force_pass in (False, True):
        if force_pass:
            os.environ["SPHFLUID_SLAB_COUNT_PASS"] = "1"
        try:
            with native.Context(device=0) as ctx:
                ctx.set_boundary(sc.bpos, sc.bparams)
                ctx.slab_upload(sc.pos, sc.vel, idx, ids, mats, sc.grid_size, capacity=2 * sc.n)
                left = torch.zeros((sc.n, 8), dtype=torch.float32, device="cuda:0")
                right = torch.zeros_like(left)
                log = []
                for lo, hi in cuts:
                    log.append(tuple(ctx.slab_pack(lo, hi, 2.0, left.data_ptr(), right.data_ptr(), sc.n)))
                    ctx.step_resident(p, 1)
                    log.append(tuple(sorted(ctx.slab_counts().items())))
                runs.append((log, ctx.slab_download()))
        finally:
            os.environ.pop("SPHFLUID_SLAB_COUNT_PASS", None)
    assert runs[0][0] == runs[1][0]
    assert any(c[0] > 0 and c[2] > 0 for c in runs[0][0][::2])   # records were sent and some moved
    for a, b in zip(runs[0][1], runs[1][1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_native_rccl_exchange_between_gpus(world, tmp_path):
    """sph_slab_exchange between real devices: one rank per GPU, records over the library's own
    RCCL communicator (ncclSend / ncclRecv over xGMI), against the undecomposed run.  Needs `world`
    GPUs (RCCL refuses two ranks on one device): skipped on the one-GPU test box, there for the
    first multi-GPU node this suite meets."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    v = run_ranks("hip_native", world, 4, tmp_path)
    assert v["ok"], v
    assert v["gathered"] == v["n"] == v["unique_ids"]
    assert all(s > 0 for s in v["sent_records"])
    assert v["max_pos_err"] <= 1e-5 and v["max_vel_err"] <= 1e-5 and v["max_rho_err"] <= 1e-5, v


@pytest.mark.gpu
def test_bench_two_ranks_with_its_own_sizing(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver calls it (no launcher): it starts its two
    ranks itself, sizes the exchange buffers and the slab capacity with its own formula and runs
    40 steps of the C2 block pair while the fluid piles up against the far wall; over gloo, both
    ranks on device 0 (the box has one GPU).  The line must say 2 GPUs and stay finite."""
    env = dict(os.environ, SPH_BENCH_COMM="gloo", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C2",
                        "--steps", "30", "--warmup", "10"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["finite"] is True
    assert line["config"]["comm_backend"] == "gloo"
    assert line["slab_rank0"]["sent_records_per_step"] > 0
    assert line["value"] > 0 and "cpu_baseline" not in line


@pytest.mark.gpu
def test_bench_two_ranks_flow_across_the_cut(tmp_path):
    """`bench.py --gpus 2 --layout x`: the scene itself (C2, water in the left third) split into two
    column ranges, dam break along x across the cut, cuts moving every step (strong scaling, the
    SURVEY 8d C4 layout at a size the test box steps in seconds); over gloo on device 0"""
    env = dict(os.environ, SPH_BENCH_COMM="gloo", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C2",
                        "--layout", "x", "--steps", "30", "--warmup", "30"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["finite"] is True and line["scaling"] == "strong"
    assert line["slab_rank0"]["sent_records_per_step"] > 0
    assert line["slab_rank0"]["owned"] + line["slab_rank0"]["ghosts"] > 0
    # value counts the scene's particles once, not per rank
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 64 * 128 * 128) < 1.0


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus_over_rccl():
    """with the RCCL backend one GPU per rank is needed: `--gpus 2` on a one-GPU box fails loudly
    instead of reporting n_gpus = 1"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two GPUs")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "SPH_BENCH_COMM"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stdout + r.stderr)
