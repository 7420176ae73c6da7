#!/bin/bash
# experiment build: the sphere walk shared out among the lanes of a wave (rt_scan.hpp, RT_SPHERES_SHARED_WALK): tools/libray_amd_sw.so
cd "$(dirname "$0")/../ray_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -shared -DRT_SPHERES_SHARED_WALK "$@" -o ../../tools/libray_amd_sw.so ray_amd.hip
