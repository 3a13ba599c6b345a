#!/bin/bash
# An A/B build of libsvo_hip.so: sparse_align.hip compiled again with extra flags, linked with the other objects of the tree.
#   scripts/build_variant.sh <name> "<flags>"   ->  build/libsvo_hip_<name>.so   (run a bench with SVOH_LIB=build/libsvo_hip_<name>.so)
# build/ is git-ignored; delete the variants once their numbers are recorded (they ship to the GPU box while they exist).
set -e
cd "$(dirname "$0")/../svo_pro_universal_amd/csrc"
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $2 -c sparse_align.hip -o ../../build/sparse_align_$1.o
objs=$(ls *.o | grep -v hooks | grep -v '^sparse_align.o$')
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o ../../build/libsvo_hip_$1.so ../../build/sparse_align_$1.o $objs
echo built build/libsvo_hip_$1.so
