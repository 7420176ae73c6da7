#!/bin/bash
# experiment build: the triangle group's filter data staged through LDS instead of read through the scalar cache
cd "$(dirname "$0")/../ray_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -shared -DRT_EXPERIMENT_TRI_STAGED -o ../../tools/exp_tri_staged.so ray_amd.hip
