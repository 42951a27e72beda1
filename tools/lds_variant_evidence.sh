#!/bin/bash
# north_star's "LDS staging of hot sphere tiles", measured: commit 2446811 of this repository carries the flat-list
# kernel twice — sphere records through the scalar cache (default) and staged by the workgroup through LDS tiles
# (RAYZ_FEED=lds) — with bit-identical images.  This script (GPU box, repo root) runs that commit's bench on
# BASELINE config 3 at 128 spp for both feeds, under rocprofv3 --kernel-trace --stats, into gpurun_out/lds_variant/.
# The commit is unpacked and built beforehand on the dev box:  git archive 2446811 | tar -x -C variants/lds_2446811
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/lds_variant; mkdir -p $out
src=$GRAFT_REPO_ROOT/variants/lds_2446811
cd /tmp && export TMPDIR=/tmp
for feed in scalar lds; do
  export RAYZ_FEED=$feed
  (cd $src && timeout -k 5 200 python3 bench.py --spp 128 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_$feed.json 2> $out/bench_$feed.err); echo "bench $feed rc=$?"
  (cd $src && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$feed -o $feed -- python3 bench.py --spp 128 --steps 2 --warmup 1 --no-cpu-baseline > $out/stats_$feed.log 2>&1); echo "stats $feed rc=$?"
done
python3 - <<PY
import json
for f in ("scalar","lds"):
    d=json.load(open("$out/bench_%s.json"%f)); print(f, d["value"], "Msamples/s", d["ms_per_step"], "ms/step")
PY
