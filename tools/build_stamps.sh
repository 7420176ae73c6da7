#!/bin/bash
# diagnostic build of the library with in-kernel cycle stamps (tools/stamps.py reads it)
cd "$(dirname "$0")/../ray_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -shared -DRT_STAMPS -o ../../tools/libray_amd_stamps.so ray_amd.hip
