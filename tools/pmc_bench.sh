#!/bin/bash
# rocprofv3 evidence for profiles/rNN (GPU box, repo root): kernel stats + PMC passes of `python bench.py` (BASELINE config 3,
# flat hit list, f32) and of the BVH kernel on the same frame.  One counter group per run (gpurun refuses --pmc with traces).
#   bash tools/pmc_bench.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also"
run() { name=$1; shift; echo "== $name: $*"; timeout -k 5 400 rocprofv3 "$@" --output-format csv -d $out/$name -o $name -- $B --steps 1 --warmup 0 > $out/$name.log 2>&1; echo "$name rc=$?"; }
echo "== stats"; timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- $B --steps 2 --warmup 1 > $out/stats.log 2>&1; echo "stats rc=$?"; grep -h "^{" $out/stats.log > $out/stats_bench_line.json
run pmc_fetch --pmc FETCH_SIZE
run pmc_fetch2 --pmc FETCH_SIZE   # (bimodal from launch to launch: taken four times, the summary keeps the largest and lists all)
run pmc_fetch3 --pmc FETCH_SIZE
run pmc_fetch4 --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS
run pmc_l2 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_WAIT_ANY SQ_WAIT_INST_ANY
# the BVH kernel on the same frame (traversal bvh)
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --traversal bvh"
echo "== bvh stats"; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bvh_stats -o bvh_stats -- $B --steps 2 --warmup 1 > $out/bvh_stats.log 2>&1; echo "bvh stats rc=$?"; grep -h "^{" $out/bvh_stats.log > $out/bvh_stats_bench_line.json
run bvh_pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS
run bvh_pmc_wait --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run bvh_pmc_mem --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
run bvh_pmc_fetch --pmc FETCH_SIZE
run bvh_pmc_write --pmc WRITE_SIZE
run bvh_pmc_ta --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
