#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun): kernel trace + PMC passes, each in a run of its own.
#   tools/prof.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/ (+ summary.txt, pmc.json)
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu --no-also"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B "$@" > $out/bench_trace.log 2>&1 || { echo trace failed; tail -5 $out/bench_trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FLOPS_FP64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/pmc1 -- $B "$@" > $out/bench_pmc1.log 2>&1 || { echo pmc1 failed; tail -5 $out/bench_pmc1.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 --output-format csv -d $out/pmc2 -- $B "$@" > $out/bench_pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $out/bench_pmc2.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc3 -- $B "$@" > $out/bench_pmc3.log 2>&1 || { echo pmc3 failed; tail -5 $out/bench_pmc3.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc4 -- $B "$@" > $out/bench_pmc4.log 2>&1 || { echo pmc4 failed; tail -5 $out/bench_pmc4.log; exit 1; }
python3 $root/tools/prof_summary.py $out
