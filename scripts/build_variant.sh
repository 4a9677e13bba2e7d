#!/bin/bash
# A build of libslimm_hip.so with edits applied to a copy of the sources, for tuning experiments on the GPU box:
#   scripts/build_variant.sh NAME 'sed-expression' ['sed-expression' ...]   ->  build/var/NAME/libslimm_hip.so
# run with SLIMM_HIP_LIB=build/var/NAME/libslimm_hip.so python bench.py ...
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build/var/$NAME
rm -rf "$D"; mkdir -p "$D/csrc"
cp -r $R/slimm_amd/csrc/*.hip $R/slimm_amd/csrc/*.h $R/slimm_amd/csrc/*.inc $R/slimm_amd/csrc/*.hpp $R/slimm_amd/csrc/*.cpp $R/slimm_amd/csrc/Makefile "$D/csrc/"
cp -r $R/slimm_amd/csrc/host "$D/csrc/"; mkdir -p $R/build/var/include; cp $R/include/*.h $R/build/var/include/
for e in "$@"; do sed -i -E "$e" $D/csrc/*.hip $D/csrc/*.h; done
make -s -C "$D/csrc" ../libslimm_hip.so 2>&1 | grep -E "error|Error" || true
ls -la "$D/libslimm_hip.so"
