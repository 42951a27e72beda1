#!/bin/bash
# PMC passes over the persistent BVH kernel (config 3, 256 spp, two renders); one counter group per run.
# usage (GPU box, repo root): bash tools/pmc_bvh.sh <outdir under gpurun_out> [traversal 1|3]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; trav=${2:-1}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 $GRAFT_REPO_ROOT/tools/wf_one.py 256 $trav > $out/$name.log 2>&1; echo "$name rc=$?"; }
timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 $GRAFT_REPO_ROOT/tools/wf_one.py 256 $trav > $out/stats.log 2>&1; echo "stats rc=$?"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LEVEL_WAVES
