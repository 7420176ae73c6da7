#!/bin/bash
# Round-2 evidence pass (GPU box): PMC + trace of the bench default (c3), kernel trace of the other workloads,
# full-frame verification of every workload. Outputs under gpurun_out/.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r2}
$root/tools/prof.sh ${tag}_c3 --steps 3 --warmup 1 > $root/gpurun_out/prof_${tag}_c3.log 2>&1 || { echo prof c3 failed; tail -5 $root/gpurun_out/prof_${tag}_c3.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for w in c2 box c4 c5; do
  out=$root/gpurun_out/prof_${tag}_$w; mkdir -p $out
  st=3; [ $w = c5 ] && st=1
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu --no-also --workload $w --steps $st --warmup 1 > $out/bench_trace.log 2>&1 || { echo trace $w failed; tail -5 $out/bench_trace.log; exit 1; }
  python3 $root/tools/prof_summary.py $out > /dev/null
done
echo profile pass done
