#!/bin/bash
# where the waves of a workload wait: issue-cycle counters per instruction class and vector-memory (TA / TCP) occupancy, one PMC pass each
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  out=$root/gpurun_out/prof_waits_$w; mkdir -p $out
  B="python3 $root/bench.py --no-cpu --no-also --workload $w --steps 2 --warmup 1"
  timeout -s KILL 200 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $out/pmc1 -- $B > $out/log1.txt 2>&1 || { echo waits1 $w failed; tail -3 $out/log1.txt; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- $B > $out/log2.txt 2>&1 || { echo waits2 $w failed; tail -3 $out/log2.txt; exit 1; }
  python3 $root/tools/prof_summary.py $out | grep -v "^==" | sed "s/^/$w /"
done
