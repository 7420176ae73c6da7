#!/bin/bash
# PMC passes over the C2 frame with and without block-sorted shading (GPU box). Output: gpurun_out/r3/sort_pmc_<mode>/
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  out=$root/gpurun_out/r3/sort_pmc_$mode; mkdir -p $out
  B="python3 $root/bench.py --no-cpu --no-also --no-shares --workload c2 --steps 3 --warmup 1 --shading-sort $mode"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/bench_trace.log 2>&1 || { echo trace failed; tail -5 $out/bench_trace.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FLOPS_FP64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/pmc1 -- $B > $out/bench_pmc1.log 2>&1 || { echo pmc1 failed; tail -5 $out/bench_pmc1.log; exit 1; }
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 --output-format csv -d $out/pmc2 -- $B > $out/bench_pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $out/bench_pmc2.log; exit 1; }
  python3 $root/tools/prof_summary.py $out > $out/summary_stdout.txt 2>&1
done
echo done
