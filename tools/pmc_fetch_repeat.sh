#!/bin/bash
# FETCH_SIZE of the flat-list trace kernel, several independent passes (it is bimodal: ~9.7 MiB or ~87.6 MiB per launch).
#   bash tools/pmc_fetch_repeat.sh <outdir under gpurun_out> [n]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; n=${2:-4}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $n); do
  timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f$i -o f$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --steps 1 --warmup 0 > $out/f$i.log 2>&1
  python3 - <<PY
import csv,glob
tot=0
for f in glob.glob("$out/f$i/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": tot+=float(r["Counter_Value"])
print("pass $i FETCH_SIZE KiB", tot)
PY
done
