#!/bin/bash
# Host-side AddressSanitizer pass over the C++ callers (GPU box): tools/asan/ holds the library built with
#   hipcc ... -Xarch_host -fsanitize=address (host code only; the device code is not instrumented)
# and the six callers of ray_amd/host (abi_tour.cpp among them: every entry point of the C ABI) built with
#   clang++ -std=c++17 -O1 -g -fsanitize=address -I ray_amd/host -I include ray_amd/host/<name>.cpp -L tools/asan -lray_amd -Wl,-rpath,'$ORIGIN'
# against it. In the box's scratch copy of the tree
# the callers take the place of the product's, and the GPU tests that start them run as they are.
# (.gpurunignore lists ./tools/asan so that ordinary GPU calls do not carry the 40 MB: take that line out for the call that runs this.)
#   bash tools/asan_run.sh <log> [tools/asan | tools/ubsan]      (tools/ubsan: the same with -fsanitize=undefined -fno-sanitize-recover=undefined)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root" || exit 1
san=${2:-tools/asan}
for n in render_demo rayview_tick mirror_kat render_multi tick_bench abi_tour; do cp $san/$n ray_amd/host/$n || exit 1; done
export LD_LIBRARY_PATH=$root/$san${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}
export ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:halt_on_error=1:exitcode=77
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu --timeout 300 --timeout-method thread \
    -k "cpp or renderOneRay or shape_hit_on_the_device or multi_device or reference_signature or tick or abi_tour" > "$1" 2>&1
rc=$?
echo "pytest rc $rc" >> "$1"
ldd ray_amd/host/render_demo | grep ray_amd >> "$1"
timeout -k 10 120 ray_amd/host/tick_bench >> "$1" 2>&1; echo "tick_bench rc $?" >> "$1"
exit $rc
