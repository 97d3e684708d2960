#!/usr/bin/env python3
"""SURVEY 8(d) CPU baseline rows: the oracle (C restatement of the reference's Burst job path:
serial multi-hash-map build + OpenMP parallel-for in batches of 12, as FluidSPHByCPUSystem.cs:243)
timed on this host at 1 thread and at the job's core share, on C1, C2 (and C3 once).
It is a C restatement, never "Burst".  usage: python tools/cpu_rows.py [--c3] > out.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsimulation_amd import scene  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run(sc, threads, reps):
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        O.sph_step(sc.pos, sc.vel, sc.inv_mass, sc.params, sc.bpos, sc.bparams, sc.grid_size,
                   nthreads=threads)
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    O.build()
    share = min(O.max_threads(), int(os.environ.get("OMP_NUM_THREADS", "16")))
    print(f"# oracle/sph_oracle.c, gcc -O2 -ffp-contract=off -fopenmp, host: {os.cpu_count()} logical "
          f"CPUs, job share {share} threads; best of N runs of one OnUpdate (sph_step)")
    print("# config  fluid  boundary  threads  seconds/step  particle-updates/s")
    cfgs = ["C1", "C2"] + (["C3"] if "--c3" in sys.argv else [])
    for name in cfgs:
        sc = scene.make_config(name)
        for th in ((1, share) if name != "C3" else (share,)):
            reps = 5 if name == "C1" else (2 if name == "C2" else 1)
            dt = run(sc, th, reps)
            print(f"{name:4s} {sc.n:9d} {sc.b:8d} {th:4d} {dt:12.4f} {sc.n / dt:12.4e}", flush=True)


if __name__ == "__main__":
    main()
