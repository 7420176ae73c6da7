#!/bin/bash
# scalar data cache behaviour of the render kernel (the filter arrays reach the VALU through s_load): one PMC pass per workload
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for w in "$@"; do
  out=$root/gpurun_out/prof_sqc_$w; mkdir -p $out
  timeout -s KILL 200 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc1 -- python3 $root/bench.py --no-cpu --no-also --workload $w --steps 2 --warmup 1 > $out/bench.log 2>&1 || { echo sqc $w failed; tail -3 $out/bench.log; exit 1; }
  python3 $root/tools/prof_summary.py $out | grep -v "^==" | sed "s/^/$w /"
done
