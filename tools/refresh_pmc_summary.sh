#!/bin/bash
# Replace profiles/rNN's rocprofv3 evidence with a new tools/pmc_bench.sh output directory, keeping the history of the
# FETCH_SIZE / WRITE_SIZE values seen (pmc_summary.json: FETCH_SIZE_KiB_all_passes) and FETCH_SIZE_KiB at the larger mode.
#   bash tools/refresh_pmc_summary.sh gpurun_out/r03l profiles/r03
set -e
src=$1; dst=$2
test -f $src/stats/stats_kernel_stats.csv
python3 - "$dst" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + '/pmc_summary.json'))
json.dump({'all': d['FETCH_SIZE_KiB_all_passes'], 'mode': d['FETCH_SIZE_KiB']}, open('/tmp/_fetch_hist.json', 'w'))
PY
for n in bvh_pmc_fetch bvh_pmc_mem bvh_pmc_sq bvh_pmc_ta bvh_pmc_wait bvh_pmc_write bvh_stats pmc_fetch pmc_l2 pmc_sq pmc_write stats; do rm -rf $dst/$n; done
python3 tools/make_pmc_summary.py $src $dst > /dev/null
python3 - "$dst" <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
p = sys.argv[1] + '/pmc_summary.json'
d = json.load(open(p))
h = json.load(open('/tmp/_fetch_hist.json'))
a = h['all']
a['earlier_sets_this_round'] = [a['this_set']] + a['earlier_sets_this_round']
a['this_set'] = d['FETCH_SIZE_KiB']
a['WRITE_SIZE_KiB_seen'] = a.get('WRITE_SIZE_KiB_seen', []) + [d['WRITE_SIZE_KiB']]
d['FETCH_SIZE_KiB_all_passes'] = a
d['FETCH_SIZE_KiB'] = max(h['mode'], d['FETCH_SIZE_KiB'])
json.dump(d, open(p, 'w'), indent=1)
print('hash matches the build:', bench.kernel_sources_sha256() == d['kernel_sources_sha256'])
print('this set: FETCH', a['this_set'], 'WRITE', d['WRITE_SIZE_KiB'])
for k, v in d['kernel_stats'].items(): print(k, round(v['average_ms'], 1), 'ms', round(v['min_ms'], 1), '-', round(v['max_ms'], 1))
print(d['derived']['bvh'])
PY
