#!/bin/bash
# a variant of libsvo_hip.so with extra compile flags, for A/B runs: scripts/build_variant.sh <name> <flags...> -> build/libsvo_hip_<name>.so
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
T=/tmp/variant_$NAME
rm -rf $T && mkdir -p $T/svo_pro_universal_amd && cp -r $ROOT/include $T/ && cp -r $ROOT/svo_pro_universal_amd/csrc $T/svo_pro_universal_amd/
cd $T/svo_pro_universal_amd/csrc && rm -f *.o libsvo_hip.so && make -s EXTRA="$*" libsvo_hip.so
mkdir -p $ROOT/build && cp libsvo_hip.so $ROOT/build/libsvo_hip_$NAME.so
echo built build/libsvo_hip_$NAME.so
