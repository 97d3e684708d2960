# times density/force per launch for experimental builds of the library (SPHFLUID_LIB)
# usage: tools/variants.sh lib1.so lib2.so ...   (rest state: 10 steps; settled: steps 100..150)
for lib in "$@"; do
  for mode in "3 10" "100 50"; do
    set -- $mode
    echo "lib=$lib warmup=$1 steps=$2"
    SPHFLUID_LIB=$PWD/$lib timeout -k 10 400 python -u bench.py --config C3 --steps $2 --warmup $1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items() if k in ('density','force')})" || exit 1
  done
done
