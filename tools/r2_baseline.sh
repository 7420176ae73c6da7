#!/bin/bash
# Round-2 measurement pass (run on the GPU box through gpurun): issue-rate microbenchmark alone and under PMC,
# section shares from the stamped build, kernel traces of every workload. Outputs under gpurun_out/r2/<tag>/.
set -u
tag=${1:-base}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $root/tools/valu_bench > $out/valu_bench.txt 2>&1 || { echo valu_bench failed; tail -3 $out/valu_bench.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FLOPS_FP64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d $out/valu_pmc -- $root/tools/valu_bench > $out/valu_pmc.log 2>&1 || { echo valu pmc failed; tail -3 $out/valu_pmc.log; exit 1; }
for w in c3 c2 c4 box; do
  timeout -k 10 200 python3 $root/tools/stamps.py $w 8 > $out/stamps_$w.txt 2>&1 || { echo stamps $w failed; tail -3 $out/stamps_$w.txt; exit 1; }
done
echo baseline pass done
