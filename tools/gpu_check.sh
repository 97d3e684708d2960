#!/bin/bash
# usage (GPU box): bash tools/gpu_check.sh <tag>  -- GPU tests, then the C3 bench at rest (3+10 steps) and collapsed (50+100)
TAG=${1:-x}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_$TAG.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/t_$TAG.log
[ $rc -eq 0 ] || exit $rc
for ws in "3 10" "50 100"; do
  set -- $ws
  timeout -k 10 300 python -u bench.py --config C3 --steps $2 --warmup $1 --no-cpu-baseline --no-host-round-trip 2> gpurun_out/b_${TAG}_$1_$2.err | tee gpurun_out/b_${TAG}_$1_$2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1+$2', round(d['ms_per_step'],3), '%.3g' % d['value'], {k: round(v,3) for k,v in d['stage_ms'].items()})" || exit 1
done
