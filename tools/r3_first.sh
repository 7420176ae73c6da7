#!/bin/bash
# round 3, first GPU call: baseline tests, bench with shares, stamps of the C3 shares
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3; mkdir -p $out
cd $root
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/pytest0.log 2>&1 || { tail -20 $out/pytest0.log; exit 1; }
tail -2 $out/pytest0.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > $out/bench0.json 2> $out/bench0.err || { tail -5 $out/bench0.err; exit 1; }
cat $out/bench0.json
for spp in 256 32 16; do
  timeout -k 10 120 python tools/stamps.py c3 $spp > $out/stamps_c3_$spp.txt 2>&1 || { tail -5 $out/stamps_c3_$spp.txt; exit 1; }
  cat $out/stamps_c3_$spp.txt
done
