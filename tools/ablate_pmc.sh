# VALU instruction counts of the neighbour kernels with parts of them switched off (SPH_ABLATE)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 0 2 1; do
  export SPH_ABLATE=$a
  OUT=$R/gpurun_out/ablate_pmc_$a
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT -- python3 $R/bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1 || { echo failed; tail -5 $OUT.log; exit 1; }
  echo "ablate=$a"; python3 $R/tools/prof_summary.py $R/gpurun_out/ablate_pmc_$a/.. 2>/dev/null | grep -A0 "nothing" ; python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_neighbors" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc: print(k, {c: "%.4g"%(sum(v)/len(v)) for c,v in acc[k].items()})
PY
done
