#!/bin/bash
# Round-3 evidence pass (GPU box): kernel trace + the four PMC passes (incl. FETCH_SIZE / WRITE_SIZE) of c3, box and c5, kernel
# trace of c2 and c4, section stamps of c3 / c2 / c4 with the current build. Outputs under gpurun_out/prof_<tag>_<workload>/.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r3}
for w in c3 box c5; do
  st=3; [ $w = c5 ] && st=1
  $root/tools/prof.sh ${tag}_$w --workload $w --steps $st --warmup 1 --no-shares > $root/gpurun_out/prof_${tag}_$w.log 2>&1 || { echo prof $w failed; tail -5 $root/gpurun_out/prof_${tag}_$w.log; exit 1; }
  echo "prof $w done"
done
cd /tmp && export TMPDIR=/tmp
for w in c2 c4; do
  out=$root/gpurun_out/prof_${tag}_$w; mkdir -p $out
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --no-cpu --no-also --no-shares --workload $w --steps 3 --warmup 1 > $out/bench_trace.log 2>&1 || { echo trace $w failed; tail -5 $out/bench_trace.log; exit 1; }
  python3 $root/tools/prof_summary.py $out > /dev/null
  echo "trace $w done"
done
cd $root
for w in c3 c2 c4; do
  timeout -k 10 200 python tools/stamps.py $w 64 > gpurun_out/${tag}_stamps_${w}_spp64.txt 2>&1 || { echo stamps $w failed; exit 1; }
done
echo profile pass done
