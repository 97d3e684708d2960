#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the SPH step on MI355X, with roofline and CPU baseline.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 launched by torch.distributed.run,
one rank per GPU) prints ONE JSON line on rank 0.  A step = one whole SPH step (keys, radix sort,
cell table, reorder, density/pressure, force, apply + integrate + colliders) of the device-resident
state; inputs are in HBM when the timed region starts.  Default workload: C3, the 16 M-particle
dam break BASELINE.json's single-GPU target is quoted on; synthetic data.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fluidsimulation_amd import native, scene  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# algorithmic bytes per fluid particle per launch of each stage (the sort of the
# step entry points is priced where `alg` is built, further down).
ALG_BYTES = {"keys": 20, "table": 6, "reorder": 52, "density": 20, "force": 44, "integrate": 68}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 50 warm-up + 200 timed
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="C3", choices=list(scene.CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-round-trip", action="store_true")
    ap.add_argument("--no-stage-breakdown", action="store_true",
                    help="skip the second pass that times the stages other than the neighbour kernels")
    ap.add_argument("--layout", default=None, choices=["z", "x"],
                    help="N > 1 only.  z (default but for C4): weak scaling, every rank owns one --config "
                         "sized block, flow along z, balanced slabs, fixed cuts.  x (default for C4, SURVEY "
                         "8d): strong scaling, the --config scene itself split into N column ranges, dam "
                         "break along x ACROSS the cuts, which move with the flow")
    ap.add_argument("--cpu-baseline-config", default=None,
                    help="workload of the bounded CPU sample (default: same as --config)")
    return ap.parse_args()


def cpu_baseline(sc, name):
    """The oracle (C restatement of the reference's Burst job path) timed on this host: one
    FluidSPHByCPUSystem.OnUpdate of the same workload.  Reported, never the target."""
    from oracle import oracle as O
    O.build()
    # the GPU box gives one job 16 of the machine's cores (OMP_NUM_THREADS is set to that); nproc
    # there reports the whole machine
    threads = min(O.max_threads(), int(os.environ.get("OMP_NUM_THREADS", "16")))
    t0 = time.perf_counter()
    O.sph_step(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.bpos, sc.bparams, sc.grid_size,
               nthreads=threads)
    dt = time.perf_counter() - t0
    return {"value": sc.n / dt, "unit": "particle-updates/s", "cores": threads, "kind": "port",
            "sample": f"1 step of {name} ({sc.n} fluid + {sc.b} boundary particles), C restatement "
                      f"of the reference's CPU path (serial map build + parallel-for jobs), "
                      f"{dt:.2f} s"}


def measured_traffic(kernel, config, n):
    """HBM bytes per launch from the committed PMC profile (profiles/hbm_traffic.json), or None
    when there is no measurement for this kernel and workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f)
        if t["config"] != config or t["n"] != n or kernel not in t:
            return None
        k = t[kernel]
        return (2.0 * k["fetch_size_kb"] + k["write_size_kb"]) * 1024.0
    except (OSError, ValueError, KeyError):
        return None


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes
    (torch.distributed.run, one per GPU) before anything in this process has touched the GPU,
    relay their output (rank 0 prints the JSON line) and exit with their status."""
    import socket
    import subprocess
    import torch
    comm = os.environ.get("SPH_BENCH_COMM", "nccl")
    ngpu = torch.cuda.device_count()      # counting devices does not initialise the GPU
    if comm == "nccl" and ngpu < a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but {ngpu} GPU(s) visible; RCCL needs one GPU per "
                 f"rank (SPH_BENCH_COMM=gloo rehearses the exchange on fewer GPUs than ranks)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # SPH_BENCH_COMM=gloo rehearses the multi-rank path on fewer GPUs than ranks
        comm = os.environ.get("SPH_BENCH_COMM", "nccl")
        ngpu = torch.cuda.device_count()
        local = local % max(ngpu, 1)
        torch.cuda.set_device(local)
        if comm == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(comm)

    sim = None
    scaling = "weak"
    if world == 1:
        sc = scene.make_config(a.config, jittered=True)
        p = native.make_params(sc.grid_size, body_bounds=(sc.tank_min, sc.tank_max))
        ctx = native.Context(device=local, timings=2)   # events around the two neighbour kernels only
        ctx.set_boundary(sc.bpos, sc.bparams)
        ctx.set_colliders(sc.boxes)
        ctx.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
        n_local = sc.n
        parallelism = "1 gpu"
        workload = (f"{a.config} dam break: {sc.n} fluid + {sc.b} boundary particles + "
                    f"{len(sc.boxes)} box colliders, dt=1/60, h=1.0, jitter 0.02")
    else:
        from fluidsimulation_amd import slab
        nx, ny, nz, ci = scene.CONFIGS[a.config]
        layout = a.layout or ("x" if a.config == "C4" else "z")
        sp = 2.0 * float(scene.RADIUS)
        if layout == "z":
            # weak scaling: every rank owns one --config sized block of a dam break that is `world`
            # blocks wide in x and flows along z (every x column carries the same flow, so the slabs
            # stay balanced); x-slab decomposition, halo 2 h, fixed cuts
            sc, ids = scene.dam_break_slab(nx * world, ny, nz, rank * nx, (rank + 1) * nx,
                                           jitter_seed=scene.JITTER_SEED_BASE + ci, name=a.config)
            edges = [-np.inf] + [r * nx * sp for r in range(1, world)] + [np.inf]
            sim_kw = {}
            scaling = "weak"
        else:
            # strong scaling: the scene itself, water in the left third of the tank,
            # split into `world` column ranges; the dam breaks along x, across the cuts, which move
            # towards equal counts every step (at most one halo a step)
            cols = [nx * r // world for r in range(world + 1)]
            sc, ids = scene.dam_break_columns(nx, ny, nz, cols[rank], cols[rank + 1],
                                              jitter_seed=scene.JITTER_SEED_BASE + ci, name=a.config)
            edges = [-np.inf] + [cols[r] * sp for r in range(1, world)] + [np.inf]
            sim_kw = dict(rebalance_every=1, domain=(0.0, 3.0 * nx * sp))
            scaling = "strong"
        p = native.make_params(sc.grid_size, body_bounds=(sc.tank_min, sc.tank_max))
        table, index = slab.material_table(sc.inv_mass, sc.params)
        ctx = native.Context(device=local, timings=2)   # events around the two neighbour kernels only
        ctx.set_boundary(sc.bpos, sc.bparams)
        ctx.set_colliders(sc.boxes)
        face = ny * nz * 4            # lattice sites within the 2 h halo of one face at t = 0
        # the fluid piles up against the far wall while it runs: room for 3x the rest-state halo
        # population in every exchange buffer, and for two generations of ghosts from both sides
        # (with moving cuts a rank may also come to own more than it started with)
        buf = 12 * face
        ctx.slab_upload(sc.pos, sc.vel, index, ids, table, sc.grid_size,
                        capacity=sc.n + 4 * buf + (sc.n // 2 if layout == "x" else 0))
        be = slab.HipSlabBackend(ctx, p, f"cuda:{local}", buffer_records=buf)
        # Transport of the records.  Over RCCL the default is the library's own communicator
        # (sph_slab_exchange: what a host without a communication layer, the C# host, would call);
        # SPH_BENCH_EXCHANGE=torch selects torch.distributed's send/recv of the same buffers.  The
        # gloo rehearsal (several ranks on one GPU, which RCCL refuses) always goes through torch.
        want_native = dist.get_backend() == "nccl" and os.environ.get("SPH_BENCH_EXCHANGE", "native") == "native"
        transport = f"torch.distributed {dist.get_backend()} send/recv (batch_isend_irecv)"
        if want_native:
            import torch
            err = ""
            try:
                be.init_native_exchange(rank, world)
            except Exception as e:       # every rank must agree on the transport
                err = str(e)
            ok = torch.tensor([0 if err else 1], device=f"cuda:{local}")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if bool(ok.item()):
                transport = "rccl ncclSend/ncclRecv inside libsphfluid (sph_slab_exchange, the library's own communicator)"
            else:
                if be.native_exchange:
                    ctx.slab_comm_destroy()
                    be.native_exchange = False
                print(f"bench.py rank {rank}: native RCCL exchange unavailable ({err or 'another rank failed'}); "
                      f"falling back to torch.distributed", file=sys.stderr)
                transport += " [library's own RCCL communicator failed to open]"
        sim = slab.SlabSimulation(be, edges, halo=2.0 * sc.grid_size, **sim_kw)
        n_local = sc.n
        parallelism = (f"x-slab decomposition over {world} ranks ({scaling} scaling), ghosts within 2h + "
                       f"ownership transfers every step; transport: {transport}")
        if layout == "z":
            workload = (f"{world} x {a.config} block dam break along z: {sc.n} fluid particles per "
                        f"GPU ({world * sc.n} total), {sc.b} boundary particles + {len(sc.boxes)} "
                        f"static boxes per rank, dt=1/60, h=1.0, jitter 0.02, fixed cuts")
        else:
            workload = (f"{a.config} dam break along x split into {world} column ranges with moving "
                        f"cuts: {nx * ny * nz} fluid particles in total, {sc.b} boundary particles + "
                        f"{len(sc.boxes)} static boxes on every rank, dt=1/60, h=1.0, jitter 0.02")

    def run(nsteps):
        if sim is None:
            ctx.step_resident(p, nsteps)
        else:
            sim.step(nsteps)

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    run(a.warmup)
    barrier()
    ctx.reset_timings()
    t0 = time.perf_counter()
    run(a.steps)
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = ctx.timings()
    total_particles = n_local * world
    if sim is not None and scaling == "strong":
        total_particles = scene.CONFIGS[a.config][0] * scene.CONFIGS[a.config][1] * scene.CONFIGS[a.config][2]
    if sim is None:
        pos, vel = ctx.download()
        slab_info = None
    else:
        pos, vel, _, _ = ctx.slab_download()
        slab_info = dict(ctx.slab_counts(), sent_records_per_step=sim.sent_records //
                         max(a.steps + a.warmup, 1), transfers_out=sim.transfers_out)
    finite = bool(np.isfinite(pos).all() and np.isfinite(vel).all())
    if dist is not None:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        f = torch.tensor([1 if finite else 0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        finite = bool(f.item())

    if rank == 0:
        n = n_local
        ms_step = elapsed / a.steps * 1e3
        # density and force: HIP events on the library's stream over the timed region itself
        stage_ms = {k: tm["ms"][k] / max(tm["steps"], 1) for k in tm["ms"]}
        if sim is None and not a.no_stage_breakdown:
            # the other stages: a second pass of the same run with every stage timed (the events
            # of all nine stages put a bubble between every two kernels, 1.4 % of a C3 step, which
            # the timed region should not pay)
            with native.Context(device=local, timings=1) as cb:
                cb.set_boundary(sc.bpos, sc.bparams)
                cb.set_colliders(sc.boxes)
                cb.upload(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.grid_size)
                cb.step_resident(p, a.warmup)
                cb.reset_timings()
                cb.step_resident(p, min(a.steps, 20))
                cb.synchronize()
                tb = cb.timings()
            for k in tb["ms"]:
                if k not in ("density", "force"):
                    stage_ms[k] = tb["ms"][k] / max(tb["steps"], 1)
        # sort inside the step entry points: counting sort by cell, 4 + 4 (count: key in, place out)
        # + 8 + 8 (placement: key and place in, permutation and sorted key out) bytes per particle
        # (SURVEY section 8d prices a radix sort, 4 + 16 x passes = 52: sph_radix_sort still is one)
        alg = dict(ALG_BYTES, sort=24)
        dom = max(("density", "force"), key=lambda k: stage_ms[k])
        achieved = alg[dom] * n / (stage_ms[dom] * 1e-3) / 1e9
        step_alg = sum(alg.values())
        out = {
            "metric": "particle-updates/sec per SPH step", "value": total_particles * a.steps / elapsed,
            "unit": "particle-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload + f"; steps {a.warmup}..{a.warmup + a.steps} of the collapse",
                       "particles_per_gpu": n, "parallelism": parallelism,
                       "comm_backend": (dist.get_backend() if dist is not None else "none"),
                       "exchange": ("none" if sim is None else
                                    ("libsphfluid sph_slab_exchange (RCCL)" if be.native_exchange
                                     else "torch.distributed send/recv"))},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(dom, a.config, n) if world == 1 else None,
                         "limiter": "fp32 VALU issue, not HBM: SQ_INSTS_VALU x 2.6 cycles (the measured "
                                    "issue rate, profiles/r2_microbench_valu.txt) is 0.87 of the density "
                                    "kernel's SIMD cycles and 0.81 of the force kernel's "
                                    "(profiles/r3_d_C3_rocprofv3_summary.txt); `traffic` is that profile's "
                                    "PMC figure (2 x FETCH_SIZE + WRITE_SIZE), not measured in this run",
                         "alg_bytes_per_particle": alg[dom],
                         "avg_launch_ms": stage_ms[dom]},
            # the two neighbour kernels take nearly the same time: both fractions, whichever leads
            "neighbour_kernels": {k: {"alg_bytes_per_particle": alg[k], "avg_launch_ms": stage_ms[k],
                                      "achieved_GBs": alg[k] * n / (stage_ms[k] * 1e-3) / 1e9,
                                      "frac": alg[k] * n / (stage_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                  for k in ("density", "force")},
            "step_roofline": {"alg_bytes_per_particle_step": step_alg,
                              "achieved_GBs": step_alg * n / (ms_step * 1e-3) / 1e9,
                              "frac": step_alg * n / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "stage_ms": stage_ms, "finite": finite,
        }
        if slab_info is not None:
            out["slab_rank0"] = slab_info
        if world == 1 and not a.no_host_round_trip:
            # the span FluidSPHByGPUSystem.cs:227-252 pays every frame: upload, step, download of
            # caller-owned host arrays -- reported beside `value`, never as it.  Twice: from
            # pageable memory, and with SphConfig.pinHostArrays (the arrays registered once, as a
            # host with persistent NativeArrays would run it)
            def round_trip(c2, frames):
                hp, hv = pos.copy(), vel.copy()
                c2.set_boundary(sc.bpos, sc.bparams)
                c2.set_colliders(sc.boxes)
                c2.step_inplace(p, hp, hv, sc.inv_mass, sc.params)   # first frame: allocations, pinning
                t1 = time.perf_counter()
                for _ in range(frames):
                    c2.step_inplace(p, hp, hv, sc.inv_mass, sc.params)
                dt_ = (time.perf_counter() - t1) / frames
                st = c2.pin_stats()
                if st["failed"]:
                    import resource
                    st["reason"] = c2.lib.sph_last_error(c2.h).decode()
                    st["rlimit_memlock"] = resource.getrlimit(resource.RLIMIT_MEMLOCK)[0]
                c2.unpin_host()          # hp / hv go away with this frame
                return dt_, st
            ctx.close()
            with native.Context(device=local) as c2:
                rt, _ = round_trip(c2, 3)
            with native.Context(device=local, pin_host_arrays=True) as c2:
                rtp, pin = round_trip(c2, 3)
            out["host_round_trip"] = {"ms_per_step": rt * 1e3, "value": n / rt,
                                      "ms_per_step_pinned": rtp * 1e3, "value_pinned": n / rtp,
                                      "pinned_arrays": pin,
                                      "unit": "particle-updates/s",
                                      "what": "sph_step: upload 24 B + step + download 24 B per "
                                              "particle on the caller's arrays, 3 frames after a "
                                              "first one; pageable / registered host memory"}
        if not a.no_cpu_baseline and world == 1:
            bsc = sc if not a.cpu_baseline_config else scene.make_config(a.cpu_baseline_config)
            out["cpu_baseline"] = cpu_baseline(bsc, a.cpu_baseline_config or a.config)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
