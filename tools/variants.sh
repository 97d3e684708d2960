# times density/force per launch for experimental builds of the library (SPHFLUID_LIB)
# usage: tools/variants.sh "<warmup> <steps>" lib1.so lib2.so ...
mode=$1; shift
for lib in "$@"; do
  set -- $mode
  echo "lib=$lib warmup=$1 steps=$2"
  SPHFLUID_LIB=$PWD/$lib timeout -k 10 400 python -u bench.py --config C3 --steps $2 --warmup $1 --no-cpu-baseline --no-host-round-trip | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})" || exit 1
done
