#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_two.sh <a.so> <b.so> <rounds> workload:spp ...   (best of 3 frames per run, alternating)
set -u
a=$1; b=$2; rounds=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 $rounds); do
  for w in "$@"; do
    for lib in $a $b; do
      timeout -k 10 100 python3 $root/tools/exp_bench.py $lib ${w%%:*} ${w##*:} 2>&1 | grep -v amdgpu.ids | sed "s/^/round $r /" || exit 1
    done
  done
done
