#!/bin/bash
# experiment build: every lane walks a BVH over the sphere pairs instead of the wave-level box / filter levels
cd "$(dirname "$0")/../ray_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fPIC -shared -DRT_EXPERIMENT_SPHERE_BVH -o ../../tools/exp_sphere_bvh.so ray_amd.hip
