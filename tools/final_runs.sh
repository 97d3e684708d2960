#!/bin/bash
# usage (GPU box): bash tools/final_runs.sh <tag>  -- the bench lines of every configuration, for profiles/<tag>_bench_*.json
TAG=${1:-rX}
run() { # name, args...
  name=$1; shift
  timeout -k 10 600 python -u bench.py "$@" > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err || { echo "$name failed"; tail -3 gpurun_out/${TAG}_bench_$name.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/${TAG}_bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['steps'], d['warmup'], round(d['ms_per_step'],3), '%.3g' % d['value'], {k: round(v,3) for k,v in d['stage_ms'].items()})"
}
run C3_default --config C3 || exit 1
run C3_50_100 --config C3 --steps 100 --warmup 50 --no-cpu-baseline || exit 1
run C3_driver --config C3 --steps 20 --warmup 5 --no-cpu-baseline || exit 1
run C1 --config C1 --no-cpu-baseline --no-host-round-trip || exit 1
run C2 --config C2 --no-cpu-baseline --no-host-round-trip || exit 1
run C5 --config C5 --no-cpu-baseline --no-host-round-trip || exit 1
