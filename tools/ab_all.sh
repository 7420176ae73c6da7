#!/bin/bash
# frame times of every workload with the library as built (GPU box): one JSON line each under gpurun_out/r2/ab_all/<tag>/
set -u
tag=${1:-cur}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2/ab_all/$tag
mkdir -p $out
for w in c3 c2 box c4 c5; do
  timeout -k 10 200 python3 $root/bench.py --workload $w --steps 3 --warmup 1 --no-cpu --no-also > $out/$w.json 2> $out/$w.err || { echo $w failed; tail -3 $out/$w.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$out/$w.json').read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], 'ms', d['value'], d['unit'])"
done
