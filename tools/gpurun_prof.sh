#!/bin/bash
# usage (on the GPU box): bash tools/gpurun_prof.sh <tag> <config> -- kernel-trace stats + two PMC passes
set -o pipefail
TAG=${1:-r1}; CFG=${2:-C3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 100 --warmup 50 --no-cpu-baseline --no-host-round-trip --no-stage-breakdown > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-host-round-trip --no-stage-breakdown > $OUT/pmc1.log 2>&1 || { echo pmc1 failed; tail -5 $OUT/pmc1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc2 -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-host-round-trip --no-stage-breakdown > $OUT/pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $OUT/pmc2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-host-round-trip --no-stage-breakdown > $OUT/pmc3.log 2>&1 || { echo pmc3 failed; tail -5 $OUT/pmc3.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-host-round-trip --no-stage-breakdown > $OUT/pmc4.log 2>&1 || { echo pmc4 failed; tail -5 $OUT/pmc4.log; exit 1; }
grep "^{" $OUT/trace.log | tail -1 > $OUT/bench_line.json
python3 $R/tools/prof_summary.py $OUT 100 50 > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only the small files
find $OUT -name '*.csv' -size +3M -delete
find $OUT -name '*.db' -delete
