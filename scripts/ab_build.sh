#!/bin/bash
# Builds variants of libspotlight_hip.so for same-box A/B runs: scripts/ab_build.sh <tag> <extra hipcc -D flags...>
# -> spotlight_amd/csrc/ab/libspotlight_hip_<tag>.so   (select with SPOTLIGHT_HIP_LIB=<path>)
set -e
TAG=$1; shift
cd "$(dirname "$0")/../spotlight_amd/csrc"
mkdir -p ab/$TAG
for f in *.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-function -Wno-pass-failed "$@" -c $f -o ab/$TAG/${f%.hip}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libspotlight_hip_$TAG.so ab/$TAG/*.o
rm -rf ab/$TAG
echo built ab/libspotlight_hip_$TAG.so
