#!/bin/bash
# usage (GPU box): bash tools/cap1_sweep.sh "<warmup> <steps>" cap1...  -- C3 step time against the record cap of single-cell blocks
mode=$1; shift
for cap in "$@"; do
  set -- $mode
  echo "cap1=$cap warmup=$1 steps=$2"
  SPHFLUID_CAP1=$cap timeout -k 10 400 python -u bench.py --config C3 --steps $2 --warmup $1 --no-cpu-baseline --no-host-round-trip 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})" || exit 1
done
