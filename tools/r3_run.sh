#!/bin/bash
# round 3: GPU tests, then A/B material. usage: tools/r3_run.sh <tag> [pytest -k expr]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; kexpr=${2:-}
out=$root/gpurun_out/r3; mkdir -p $out
cd $root
if [ -n "$kexpr" ]; then
  timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "$kexpr" > $out/pytest_$tag.log 2>&1 || { tail -30 $out/pytest_$tag.log; exit 1; }
else
  timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/pytest_$tag.log 2>&1 || { tail -30 $out/pytest_$tag.log; exit 1; }
fi
tail -2 $out/pytest_$tag.log
