#!/bin/bash
# Build a variant of the library for same-box A/B runs (LTPL_HIP_LIB=<path>): tools/build_variant.sh <name> [-DSWITCH ...]
#   -> graphbasedlocaltrajectoryplanner_amd/csrc/variants/<name>.so (git-ignored, travels to the GPU box with the snapshot)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/graphbasedlocaltrajectoryplanner_amd/csrc/variants
cd $ROOT/graphbasedlocaltrajectoryplanner_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 ${OPT:--O3} -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function "$@" -o variants/$NAME.so ltpl_hip.hip
echo "built variants/$NAME.so ($*)"
